// LDS-DMA staging rate per CU as a function of the access shape of one wave-instruction (1 KiB): which source
// addresses the 64 lanes present.  One workgroup per CU streams an L2-resident table into LDS over and over; nothing
// reads the LDS.  Prints GB/s per CU and bytes per clock (2.4 GHz nominal).
//   hipcc -O2 --offload-arch=gfx950 tools/probes/dma_rate.hip -o tools/probes/dma_rate.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// SHAPE 0: 1 KiB contiguous.  1: 8 rows x 128 B, row stride `stride`, chunks in order.  2: as 1 with the 16-byte chunks
// XOR-permuted inside the row.  3: 16 rows x 64 B.  4: 64 rows x 16 B (worst case).
// KIND 0: global_load_lds.  1: buffer_load ... offen lds.
template <int SHAPE, int KIND, int NW>
__global__ __launch_bounds__(NW * 64) void dma_kernel(const char* __restrict__ tab, size_t tab_bytes, int stride, int iters,
                                                      int inflight) {
  __shared__ __attribute__((aligned(1024))) char smem[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  size_t lane_off;
  int piece_adv;  // bytes from one piece to the next
  if (SHAPE == 0) { lane_off = lane * 16; piece_adv = 1024; }
  else if (SHAPE == 1) { lane_off = (size_t)(lane >> 3) * stride + (lane & 7) * 16; piece_adv = 8 * stride; }
  else if (SHAPE == 2) { const int r = lane >> 3; lane_off = (size_t)r * stride + (((lane & 7) ^ (r & 7)) * 16); piece_adv = 8 * stride; }
  else if (SHAPE == 3) { lane_off = (size_t)(lane >> 2) * stride + (lane & 3) * 16; piece_adv = 16 * stride; }
  else { lane_off = (size_t)lane * stride; piece_adv = 64 * stride; }
  const size_t span = tab_bytes / 2;  // pieces wrap inside the first half so that lane offsets stay in range
  size_t pos = ((size_t)(blockIdx.x * NW + wave) * 37 * piece_adv) % span;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tab), 0, (int)tab_bytes, 0x00020000);
  char* dst = smem + wave * (64 * 1024 / NW);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) __builtin_amdgcn_global_load_lds((gptr_t)(tab + pos + lane_off), (lptr_t)(dst + u * 1024), 16, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + u * 1024), 16, (int)lane_off, (int)pos, 0, 0);
      pos += (size_t)piece_adv * NW;
      if (pos >= span) pos -= span;
    }
    if (inflight == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (inflight == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (inflight == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the staging pattern of a phased GEMM loop: NP pieces per wave, a counted wait, NB barriers, repeat
template <int NP, int NB, int WAITN, int LGKM = 0, int SAME = 0>
__global__ __launch_bounds__(512) void dma_phase_kernel(const char* __restrict__ tab, size_t tab_bytes, int stride, int iters) {
  __shared__ __attribute__((aligned(1024))) char smem[128 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane >> 3;
  const size_t lane_off = (size_t)r * stride + (((lane & 7) ^ (r & 7)) * 16);
  const int piece_adv = 8 * stride;
  const size_t span = tab_bytes / 2;
  // SAME 1: every workgroup walks the same addresses at the same time (a weight tile shared by all CUs);
  // SAME 2: the same addresses, each workgroup starting at its own offset into the walk
  size_t pos = SAME == 1 ? ((size_t)wave * piece_adv) % span
             : SAME == 2 ? ((size_t)(wave + 8 * (blockIdx.x * 13 % 64)) * piece_adv) % span
                         : ((size_t)(blockIdx.x * 8 + wave) * 37 * piece_adv) % span;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tab), 0, (int)tab_bytes, 0x00020000);
  char* dst = smem + wave * 16384;
  int slot = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + slot * 1024), 16, (int)lane_off, (int)pos, 0, 0);
      slot = (slot + 1) & 15;
      pos += (size_t)piece_adv * 8;
      if (pos >= span) pos -= span;
    }
    if (WAITN == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (WAITN == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < NB; ++b) __builtin_amdgcn_s_barrier();
    if (LGKM == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (LGKM == 2) {  // a fragment read of data that landed long ago, waited for
      const float v = *reinterpret_cast<const volatile float*>(smem + ((slot + 4) & 15) * 1024 + wave * 16384 + lane * 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(v) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// ping-pong structure: the waves 0..3 and 4..7 alternate; a group issues NP pieces, waits, then two barriers pass
// before its next issue, and the other group issues in between
template <int NP, int FILL>
__global__ __launch_bounds__(512) void dma_pingpong_kernel(const char* __restrict__ tab, size_t tab_bytes, int stride, int iters) {
  __shared__ __attribute__((aligned(1024))) char smem[128 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane >> 3;
  const size_t lane_off = (size_t)r * stride + (((lane & 7) ^ (r & 7)) * 16);
  const int piece_adv = 8 * stride;
  const size_t span = tab_bytes / 2;
  size_t pos = ((size_t)(blockIdx.x * 8 + wave) * 37 * piece_adv) % span;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tab), 0, (int)tab_bytes, 0x00020000);
  char* dst = smem + wave * 16384;
  int slot = 0;
  float filler = (float)lane;
  if (wave >= 4) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + slot * 1024), 16, (int)lane_off, (int)pos, 0, 0);
      slot = (slot + 1) & 15;
      pos += (size_t)piece_adv * 8;
      if (pos >= span) pos -= span;
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (FILL) {  // stands for the products: FILL dependent vector instructions (4 cycles each)
#pragma unroll
      for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(filler));
    }
    __builtin_amdgcn_s_barrier();
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(filler) : "memory");
}
template <int NP, int FILL>
void run_pingpong(const char* tab, size_t tab_bytes, int stride) {
  const int iters = 8000;
  hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
  dma_pingpong_kernel<NP, FILL><<<256, 512>>>(tab, tab_bytes, stride, 50);
  HIP(hipEventRecord(e0));
  dma_pingpong_kernel<NP, FILL><<<256, 512>>>(tab, tab_bytes, stride, iters);
  HIP(hipEventRecord(e1)); HIP(hipEventSynchronize(e1));
  float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
  const double bytes_cu = (double)iters * NP * 1024 * 8;
  printf("ping-pong: %d pieces / wave / phase, %3d filler instructions: %7.1f GB/s per CU, %6.0f ns per phase\n", NP, FILL,
         bytes_cu / ms / 1e6, ms * 1e6 / iters);
}

template <int NP, int NB, int WAITN, int LGKM = 0, int SAME = 0>
void run_phase(const char* tab, size_t tab_bytes, int stride) {
  const int iters = 8000;
  hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
  dma_phase_kernel<NP, NB, WAITN, LGKM, SAME><<<256, 512>>>(tab, tab_bytes, stride, 50);
  HIP(hipEventRecord(e0));
  dma_phase_kernel<NP, NB, WAITN, LGKM, SAME><<<256, 512>>>(tab, tab_bytes, stride, iters);
  HIP(hipEventRecord(e1)); HIP(hipEventSynchronize(e1));
  float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
  const double bytes_cu = (double)iters * NP * 1024 * 8;
  printf("phased (lgkm wait %d, same-address mode %d): %d pieces / wave, vmcnt(%2d), %d barriers, table %4zu MiB: %7.1f GB/s per CU, %5.1f B/clk, %6.0f ns per iteration\n", LGKM, SAME, NP, WAITN, NB,
         tab_bytes >> 20, bytes_cu / ms / 1e6, bytes_cu / (ms * 1e-3) / 2.4e9, ms * 1e6 / iters);
}

// Does a wave that streams from HBM slow down a wave of the same CU that streams from L2?  Waves 0..3 read the hot
// table, waves 4..7 the cold one (MIX 1), or every wave alternates hot and cold pieces (MIX 2), or all hot (MIX 0).
// Each wave counts its own pieces until wave 0 of the block has done `iters`.
template <int MIX>
__global__ __launch_bounds__(512) void dma_mix_kernel(const char* __restrict__ hot, size_t hot_bytes, const char* __restrict__ cold,
                                                      size_t cold_bytes, int iters, unsigned long long* cycles) {
  __shared__ __attribute__((aligned(1024))) char smem[128 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane >> 3, stride = 512;
  const unsigned lane_off = (unsigned)r * stride + (((lane & 7) ^ (r & 7)) * 16);
  const int piece_adv = 8 * stride;
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(hot), 0, (int)hot_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(cold), 0, (int)0x7fffffff, 0x00020000);
  size_t hpos = ((size_t)(blockIdx.x * 8 + wave) * 37 * piece_adv) % (hot_bytes / 2);
  size_t cpos = ((size_t)(blockIdx.x * 8 + wave) * 4096ull * piece_adv) % (cold_bytes / 2);
  char* dst = smem + wave * 16384;
  const bool cold_wave = MIX == 1 && wave >= 4;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool c = cold_wave || (MIX == 2 && (u & 1));
      if (c) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (lptr_t)(dst + u * 1024), 16, (int)lane_off, (int)cpos, 0, 0);
        cpos += piece_adv; if (cpos >= cold_bytes / 2) cpos -= cold_bytes / 2;
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lptr_t)(dst + u * 1024), 16, (int)lane_off, (int)hpos, 0, 0);
        hpos += (size_t)piece_adv * 8; if (hpos >= hot_bytes / 2) hpos -= hot_bytes / 2;
      }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MIX>
void run_mix(const char* hot, size_t hot_bytes, const char* cold, size_t cold_bytes) {
  const int iters = 2000;
  unsigned long long* cyc; HIP(hipMalloc(&cyc, 256 * 8 * 8));
  dma_mix_kernel<MIX><<<256, 512>>>(hot, hot_bytes, cold, cold_bytes, iters, cyc);
  HIP(hipDeviceSynchronize());
  unsigned long long h[2048]; HIP(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
  double a[2] = {0, 0};
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) a[w >> 2] += (double)h[b * 8 + w] / (256 * 4);
  const double bytes = (double)iters * 4 * 1024;
  printf("mix %d: waves 0-3 %.1f B/clk per wave (%.0f cycles per piece), waves 4-7 %.1f B/clk per wave (%.0f cycles per piece)\n", MIX,
         bytes / a[0], a[0] / (iters * 4), bytes / a[1], a[1] / (iters * 4));
  (void)hipFree(cyc);
}

template <int SHAPE, int KIND, int NW>
void run(const char* name, const char* tab, size_t tab_bytes, int stride, int inflight) {
  const int iters = 2000;
  hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
  dma_kernel<SHAPE, KIND, NW><<<256, NW * 64>>>(tab, tab_bytes, stride, 50, inflight);
  HIP(hipEventRecord(e0));
  dma_kernel<SHAPE, KIND, NW><<<256, NW * 64>>>(tab, tab_bytes, stride, iters, inflight);
  HIP(hipEventRecord(e1)); HIP(hipEventSynchronize(e1));
  float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
  const double bytes_cu = (double)iters * 8 * 1024 * NW;
  printf("%-34s waves %d kind %s stride %5d inflight %2d: %7.1f GB/s per CU, %5.1f B/clk, chip %5.1f TB/s\n", name, NW,
         KIND ? "buffer" : "global", stride, inflight, bytes_cu / ms / 1e6, bytes_cu / (ms * 1e-3) / 2.4e9, bytes_cu * 256 / ms / 1e9);
}

int main() {
  const size_t tab_bytes = 2u << 20;  // 2 MiB: resident in every XCD's L2
  char* tab; HIP(hipMalloc(&tab, tab_bytes)); HIP(hipMemset(tab, 1, tab_bytes));
  run<0, 0, 4>("1 KiB contiguous", tab, tab_bytes, 0, 8);
  run<0, 0, 8>("1 KiB contiguous", tab, tab_bytes, 0, 8);
  run<0, 1, 8>("1 KiB contiguous", tab, tab_bytes, 0, 8);
  run<0, 0, 8>("1 KiB contiguous", tab, tab_bytes, 0, 32);
  run<1, 0, 8>("8 rows x 128 B", tab, tab_bytes, 512, 8);
  run<1, 1, 8>("8 rows x 128 B", tab, tab_bytes, 512, 8);
  run<1, 1, 8>("8 rows x 128 B", tab, tab_bytes, 4608, 8);
  run<1, 1, 8>("8 rows x 128 B", tab, tab_bytes, 8192, 8);
  run<2, 0, 8>("8 rows x 128 B, chunks XOR-permuted", tab, tab_bytes, 512, 8);
  run<2, 1, 8>("8 rows x 128 B, chunks XOR-permuted", tab, tab_bytes, 512, 8);
  run<2, 1, 8>("8 rows x 128 B, chunks XOR-permuted", tab, tab_bytes, 512, 32);
  run<2, 1, 4>("8 rows x 128 B, chunks XOR-permuted", tab, tab_bytes, 512, 32);
  run<2, 1, 8>("8 rows x 128 B, chunks XOR-permuted", tab, tab_bytes, 4608, 8);
  run<3, 1, 8>("16 rows x 64 B", tab, tab_bytes, 512, 8);
  run<4, 1, 8>("64 rows x 16 B", tab, tab_bytes, 512, 8);
  run_phase<2, 2, 8, 0, 1>(tab, tab_bytes, 512);
  run_phase<2, 2, 8, 0, 2>(tab, tab_bytes, 512);
  run_phase<2, 2, 8, 0, 1>(tab, tab_bytes, 4608);
  run_phase<2, 2, 8, 0, 2>(tab, tab_bytes, 4608);
  run_phase<8, 2, 8, 0, 1>(tab, tab_bytes, 4608);
  run_pingpong<2, 0>(tab, tab_bytes, 512);
  run_pingpong<2, 64>(tab, tab_bytes, 512);
  run_pingpong<1, 64>(tab, tab_bytes, 512);
  run_pingpong<4, 64>(tab, tab_bytes, 512);
  run_pingpong<2, 0>(tab, tab_bytes, 4608);
  run_phase<2, 2, 8>(tab, tab_bytes, 512);
  run_phase<2, 2, 8, 1>(tab, tab_bytes, 512);
  run_phase<2, 2, 8, 2>(tab, tab_bytes, 512);
  run_phase<2, 1, 8>(tab, tab_bytes, 512);
  run_phase<2, 0, 8>(tab, tab_bytes, 512);
  run_phase<4, 2, 8>(tab, tab_bytes, 512);
  run_phase<4, 2, 12>(tab, tab_bytes, 512);
  run_phase<8, 2, 8>(tab, tab_bytes, 512);
  const size_t big_bytes = 64u << 20;  // beyond the L2s, inside the Infinity Cache
  char* big; HIP(hipMalloc(&big, big_bytes)); HIP(hipMemset(big, 1, big_bytes));
  run_phase<2, 2, 8>(big, big_bytes, 512);
  run_phase<4, 2, 12>(big, big_bytes, 512);
  run_phase<8, 2, 8>(big, big_bytes, 512);
  run<2, 1, 8>("8 rows x 128 B, XOR, 64 MiB table", big, big_bytes, 512, 8);
  run<2, 1, 8>("8 rows x 128 B, XOR, 64 MiB table", big, big_bytes, 512, 32);
  const size_t huge_bytes = 1024u << 20;  // HBM
  char* huge; HIP(hipMalloc(&huge, huge_bytes)); HIP(hipMemset(huge, 1, huge_bytes));
  run_mix<0>(tab, tab_bytes, huge, huge_bytes);
  run_mix<1>(tab, tab_bytes, huge, huge_bytes);
  run_mix<2>(tab, tab_bytes, huge, huge_bytes);
  run_phase<2, 2, 8>(huge, huge_bytes, 512);
  run<2, 1, 8>("8 rows x 128 B, XOR, 1 GiB table", huge, huge_bytes, 512, 8);
  run<2, 1, 8>("8 rows x 128 B, XOR, 1 GiB table", huge, huge_bytes, 512, 32);
  return 0;
}
