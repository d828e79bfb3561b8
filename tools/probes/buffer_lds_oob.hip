// Probe: what does an LDS-DMA buffer load (buffer_load_dwordx4 ... lds) write for lanes whose offset lies beyond
// the descriptor's range?  The LDS is pre-filled with a pattern; lanes 0..31 load in range, lanes 32..63 out of range.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void probe(const uint32_t* src, int nbytes, uint32_t* out, int soff) {
  __shared__ __attribute__((aligned(1024))) uint32_t lds[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = 0xABABABABu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // out of range for the upper half; with soff != 0 the sentinel is chosen so that voffset + soffset wraps into range
  const int voff = lane < 32 ? lane * 16 : (soff ? (int)0xFFFFFFF0u + (lane - 32) * 0 : 0x7FFFFFF0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<uint32_t> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
  uint32_t *d, *o;
  hipMalloc(&d, 1024);
  hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  for (int soff = 0; soff <= 256; soff += 256) {
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1024, o, soff);
  hipDeviceSynchronize();
  printf("soffset %d:\n", soff);
  std::vector<uint32_t> r(256);
  hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
  printf("lane 0: %08x %08x | lane 31: %08x | lane 32 (oob): %08x %08x | lane 63 (oob): %08x\n", r[0], r[1], r[31 * 4], r[32 * 4],
         r[32 * 4 + 1], r[63 * 4 + 3]);
  int zeros = 0, pattern = 0;
  for (int i = 128; i < 256; ++i) { zeros += r[i] == 0; pattern += r[i] == 0xABABABABu; }
  printf("out-of-range half: %d words zero, %d words untouched (of 128)\n", zeros, pattern);
  }
  return 0;
}
