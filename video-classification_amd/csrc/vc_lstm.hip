// Persistent LSTM recurrence for gfx950 (MI355X).
//
// One launch runs all T time steps of one layer (both directions of a bidirectional layer side by side).
// The recurrent weight W_hh [4H][H] is sharded over workgroups by hidden unit: a workgroup owns 16 units,
// i.e. 64 gate rows, and keeps them in VGPRs as MFMA A-operand fragments for the whole sequence.  Per step
// every workgroup needs the complete h_{t-1} [B][H]; it is exchanged through global memory in 8-byte
// {tag, two bf16 values} granules written with agent-scope (write-through) atomic stores and read with
// agent-scope atomic loads: the tag is the step number, so the data is its own flag and no fence, counter
// or grid barrier is needed.  Gate pre-activations come out of the MFMA with all four gates of one
// (batch row, unit) in one lane, so the cell update is pure register math; c_t never leaves the lane.
//
// Layout of the MFMA (v_mfma_f32_16x16x32_bf16, D = A x B):
//   A rows  = gate rows of 4 units, ordered unit*4 + gate      (wave w owns units 4w..4w+3 of the 16)
//   B cols  = 16 batch rows (two tiles cover the 32 rows a workgroup handles)
//   D[row = 4*(lane>>4) + r][col = lane&15]  ->  lane holds gates r = 0..3 (i, f, g, o) of
//   unit 4w + (lane>>4) for batch row (lane&15) + 16*tile.
#include <mutex>

#include "vc_common.h"

namespace {

constexpr int LTHR = 256;
constexpr int UNITS = 16;    // hidden units per workgroup
constexpr int BCHUNK = 32;   // batch rows per workgroup
constexpr unsigned SPIN_LIMIT = 1u << 20;

typedef __attribute__((address_space(1))) unsigned long long gu64;

struct LstmDir {
  const float* gates_x;  // [T][B][4H] fp32: x W_ih^T + b_ih + b_hh
  const bf16* w_hh;      // [4H][H]
  bf16* hs;              // h_t at hs[(t*B + b)*ldh + unit]
  float* act;            // [T][B][4H] post-activation gates (saved for backward)
  float* c;              // [T][B][H]
  unsigned long long* xchg;  // [chunks][2 parities][H/2 * 32] granules
  int reverse;
};
struct LstmArgs {
  LstmDir d[2];
  int T, B, H, ldh;
  int* err;
};

__device__ __forceinline__ float sigmoid_fast(float x) { return 1.f / (1.f + __expf(-x)); }

template <int NKS>  // NKS = H / 32 k-steps of the recurrent product
__global__ __launch_bounds__(LTHR, 1) void lstm_persist_fwd_kernel(LstmArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16 sH[];  // [32][H], 16-byte chunks XOR-swizzled by row
  const int H = NKS * 32;
  const LstmDir& D = a.d[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int u0 = blockIdx.x * UNITS;
  const int b0 = blockIdx.z * BCHUNK;
  const int unit = u0 + 4 * wave + q;  // the hidden unit whose gates this lane receives
  constexpr int NCH = NKS * 4;         // 16-byte chunks per h row
  constexpr int SWZ = (NCH < 16 ? NCH : 16) - 1;

  // recurrent weights of this wave's 16 gate rows, resident in registers for the whole sequence
  bf16x8 wreg[NKS];
  {
    const int row = l15;  // row of the A tile: unit (row >> 2), gate (row & 3)
    const bf16* wrow = D.w_hh + (size_t)((row & 3) * H + u0 + 4 * wave + (row >> 2)) * H + 8 * q;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) wreg[ks] = *reinterpret_cast<const bf16x8*>(wrow + 32 * ks);
  }
  gu64* xbase = (gu64*)(D.xchg + (size_t)blockIdx.z * 2 * (H / 2) * BCHUNK);
  const int per_parity = (H / 2) * BCHUNK;

  float cst[2] = {0.f, 0.f};
  for (int ts = 0; ts < a.T; ++ts) {
    const int t = D.reverse ? a.T - 1 - ts : ts;
    // input part of the pre-activations (independent of h): issue these loads before waiting on h
    float pre[2][4];
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const int b = b0 + bt * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pre[bt][r] = b < a.B ? D.gates_x[((size_t)t * a.B + b) * 4 * H + r * H + unit] : 0.f;
    }
    if (ts > 0) {
      // gather h_{t-1}: every granule carries the step that wrote it; spin (bounded) until it is this step's
      gu64* xp = xbase + ((ts - 1) & 1) * per_parity;
      constexpr int NG = NKS * 2;  // granules per thread: (H/2 * 32) / 256
      unsigned long long g[NG];
      unsigned spins = 0;
      for (;;) {
        // all loads of a sweep are issued back to back (one memory latency per sweep, not per granule)
#pragma unroll
        for (int k = 0; k < NG; ++k)
          g[k] = __hip_atomic_load(xp + tid + LTHR * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NG; ++k) ok &= (unsigned)(g[k] >> 32) == (unsigned)ts;
        if (ok) break;
        if (++spins > SPIN_LIMIT) {  // a workgroup is missing: give up loudly instead of hanging the GPU
          if (a.err) atomicExch(a.err, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int idx = tid + LTHR * k;
        const int up = idx >> 5, b = idx & 31;  // unit pair, batch row
        const int chunk = (up >> 2) ^ (b & SWZ);
        *reinterpret_cast<unsigned*>(sH + b * H + chunk * 8 + 2 * (up & 3)) = (unsigned)g[k];
      }
    }
    __syncthreads();
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (ts > 0) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
          const int b = bt * 16 + l15;
          const bf16x8 hf = *reinterpret_cast<const bf16x8*>(sH + b * H + (((4 * ks + q) ^ (b & SWZ)) << 3));
          acc[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks], hf, acc[bt], 0, 0, 0);
        }
      }
    }
    float hval[2];
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const int b = b0 + bt * 16 + l15;
      const float ig = sigmoid_fast(pre[bt][0] + acc[bt][0]), fg = sigmoid_fast(pre[bt][1] + acc[bt][1]);
      const float gg = tanhf(pre[bt][2] + acc[bt][2]), og = sigmoid_fast(pre[bt][3] + acc[bt][3]);
      const float c = fg * cst[bt] + ig * gg;
      const float h = og * tanhf(c);
      cst[bt] = c;
      hval[bt] = b < a.B ? h : 0.f;
      if (b < a.B) {
        const size_t row = (size_t)t * a.B + b;
        D.c[row * H + unit] = c;
        D.hs[row * a.ldh + unit] = (bf16)h;
        float* ap = D.act + row * 4 * H + unit;
        ap[0] = ig;
        ap[H] = fg;
        ap[2 * H] = gg;
        ap[3 * H] = og;
      }
    }
    if (ts + 1 < a.T) {
      // publish: lanes with an even unit pack it with the odd neighbour (lane ^ 16) into one granule
      gu64* xp = xbase + (ts & 1) * per_parity;
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
        const float other = __shfl_xor(hval[bt], 16, 64);
        if ((q & 1) == 0) {
          union {
            bf16 h[2];
            unsigned u;
          } pk;
          pk.h[0] = (bf16)hval[bt];
          pk.h[1] = (bf16)other;
          const int up = unit >> 1, b = bt * 16 + l15;
          const unsigned long long g = ((unsigned long long)(unsigned)(ts + 1) << 32) | pk.u;
          __hip_atomic_store(xp + up * BCHUNK + b, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __syncthreads();  // every wave has finished reading sH before the next step overwrites it
  }
}


// ---------------------------------------------------------------------------------------------------------
// Backward recurrence (BPTT), same sharding: a workgroup owns 16 hidden units.  Per step it needs
//   dh_rec[b][j] = sum_r dgates_next[b][r] * W_hh[r][j]        (r over the 4H gate rows)
// for its 16 units j, i.e. the COMPLETE gate-gradient vector of the step processed before: 4x the data the
// forward pass exchanges, too much for 8-byte tagged granules (every small write-through store is a fabric write
// of its own).  It travels as bulk data instead: the owner of (b, unit) computes the unit's four gate gradients in
// registers (dh = dh_out + dh_rec, the carried dc, the saved activations), the workgroup collects its 32 x 64
// values in LDS and writes them as whole 128-byte lines with write-through 16-byte stores, every wave drains its
// stores, and ONE lane then raises the workgroup's flag to the step number.  A consumer polls the flags of all
// producers (one wave, one lane per flag, L1-bypassing loads), passes a barrier and fetches the vector with
// L1-bypassing 16-byte loads: no fence on either side.  Two parities of the data buffer suffice: nobody can
// publish step s+2 before everybody has read step s.
// The reduction index is ordered k = 4*unit + gate (a workgroup's 64 values are contiguous); the four waves split
// k, each holding its slice of W_hh^T as MFMA A fragments in registers, and sum their 16 x 32 partials through LDS.
// ---------------------------------------------------------------------------------------------------------
struct LstmBwdDir {
  const bf16* dh_out;   // gradient of the layer output: dh_out[(t*B + b)*ldo + unit]
  const float* act;     // [T][B][4H] post-activation gates saved by the forward pass
  const float* c;       // [T][B][H]
  const bf16* w_hh_t;   // [H][4H]: W_hh^T (columns gate-major, as W_hh's rows)
  bf16* dgates;         // [T][B][4H], gate-major like act
  unsigned char* xchg;  // per batch chunk: data [2 parities][32][4H] bf16, then 64 flag words
  int reverse;
};
struct LstmBwdArgs {
  LstmBwdDir d[2];
  int T, B, H, ldo;
  int bvs;  // log2 of the batch rows a chunk exchanges (8, 16 or 32: rows beyond B in the last chunk are zeros)
  int* err;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr int BWD_FLAG_WORDS = 64;

template <int NKS>  // NKS = H / 32 = k-steps of the recurrent product per wave (4H / 32 in all)
__global__ __launch_bounds__(LTHR, 1) void lstm_persist_bwd_kernel(LstmBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int H = NKS * 32, K = 4 * H;
  constexpr int NCHK = K / 8;  // 16-byte chunks per row of the gate-gradient vector
  bf16* sD = reinterpret_cast<bf16*>(lds_raw);                                         // [32][K], chunks swizzled by row
  float* sR = reinterpret_cast<float*>(lds_raw + (size_t)BCHUNK * K * sizeof(bf16));  // [4][16][32] partial products
  bf16* sP = reinterpret_cast<bf16*>(sR + 4 * 16 * 32);                                // [32][64] this workgroup's values
  const LstmBwdDir D = blockIdx.y ? a.d[1] : a.d[0];  // a runtime index would put the struct in scratch
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int u0 = blockIdx.x * UNITS;
  const int b0 = blockIdx.z * BCHUNK;
  const int unit = u0 + 4 * wave + q;
  const int bv = 1 << a.bvs;
  const int ntile = bv > 16 ? 2 : 1;  // batch tiles of 16 that carry data
  const int nprod = H / UNITS;        // workgroups that publish to this one (itself included)

  // this wave's slice of W_hh^T: rows = the workgroup's 16 units, k = 4*unit' + gate in [wave*H, (wave+1)*H)
  bf16x8 wreg[NKS];
  {
    const bf16* wrow = D.w_hh_t + (size_t)(u0 + l15) * K;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int ub = (wave * H + 32 * ks + 8 * q) >> 2;  // first of the two units this fragment covers
#pragma unroll
      for (int e = 0; e < 8; ++e) wreg[ks][e] = wrow[(e & 3) * H + ub + (e >> 2)];
    }
  }
  // rows that are never exchanged stay zero for the whole kernel
  for (int i = tid; i < BCHUNK * K / 8; i += LTHR) reinterpret_cast<bf16x8*>(sD)[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();

  const size_t data_bytes = (size_t)2 * BCHUNK * K * sizeof(bf16);
  unsigned char* xb = D.xchg + (size_t)blockIdx.z * (data_bytes + BWD_FLAG_WORDS * 4);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)data_bytes, 0x00020000);
  gu32* flags = (gu32*)(xb + data_bytes);
  const int parity_bytes = BCHUNK * K * (int)sizeof(bf16);
  const int nload = (bv * NCHK + LTHR - 1) / LTHR;  // 16-byte loads per thread per step

  float dcar[2] = {0.f, 0.f};  // dc flowing into the step being processed (dc_next * f_next)
  for (int ts = 0; ts < a.T; ++ts) {
    const int t = D.reverse ? ts : a.T - 1 - ts;
    const bool has_prev = ts + 1 < a.T;            // a step earlier in forward order exists
    const int tp = D.reverse ? t + 1 : t - 1;
    // everything that does not depend on the recurrence is fetched before waiting for the other workgroups
    float gi[2], gf[2], gg[2], go[2], cc[2], cp[2], dho[2];
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const int b = b0 + bt * 16 + l15;
      const bool live = b < a.B && bt < ntile;
      const size_t row = (size_t)t * a.B + b;
      const float* ap = D.act + row * K + unit;
      gi[bt] = live ? ap[0] : 0.f;
      gf[bt] = live ? ap[H] : 0.f;
      gg[bt] = live ? ap[2 * H] : 0.f;
      go[bt] = live ? ap[3 * H] : 0.f;
      cc[bt] = live ? D.c[row * H + unit] : 0.f;
      cp[bt] = (live && has_prev) ? D.c[((size_t)tp * a.B + b) * H + unit] : 0.f;
      dho[bt] = live ? (float)D.dh_out[row * a.ldo + unit] : 0.f;
    }
    float dhr[2] = {0.f, 0.f};
    if (ts > 0) {
      if (wave == 0) {  // one wave polls, one lane per producer; the flags only grow
        unsigned spins = 0;
        for (;;) {
          unsigned f = (unsigned)ts;
          if (lane < nprod) f = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__all(f >= (unsigned)ts)) break;
          if (++spins > SPIN_LIMIT) {
            if (a.err) atomicExch(a.err, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      const int pbase = ((ts - 1) & 1) * parity_bytes;
      constexpr int LB = 16;  // loads in flight per thread and batch
      for (int base = 0; base < nload; base += LB) {
        u32x4 v[LB];
#pragma unroll
        for (int k = 0; k < LB; ++k) {
          const int idx = tid + LTHR * (base + k);
          if (base + k < nload && idx < bv * NCHK)
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pbase + idx * 16, 0, 16);
        }
#pragma unroll
        for (int k = 0; k < LB; ++k) {
          const int idx = tid + LTHR * (base + k);
          if (base + k < nload && idx < bv * NCHK) {
            const int b = idx / NCHK, ch = idx - b * NCHK;
            *reinterpret_cast<u32x4*>(sD + b * K + ((ch ^ (b & 15)) << 3)) = v[k];
          }
        }
      }
      __syncthreads();
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int ch = (wave * H + 32 * ks) / 8 + q;
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
          if (bt < ntile) {
            const int b = bt * 16 + l15;
            const bf16x8 df = *reinterpret_cast<const bf16x8*>(sD + b * K + ((ch ^ (b & 15)) << 3));
            acc[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks], df, acc[bt], 0, 0, 0);
          }
        }
      }
      // D[row = 4*q + r][col = l15]: unit u0 + 4q + r, batch row l15 (+16 for the second tile)
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sR[(wave * 16 + 4 * q + r) * 32 + bt * 16 + l15] = acc[bt][r];
      __syncthreads();
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int w = 0; w < 4; ++w) dhr[bt] += sR[(w * 16 + 4 * wave + q) * 32 + bt * 16 + l15];
    }
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const float dh = dho[bt] + dhr[bt];
      const float tc = tanhf(cc[bt]);
      const float dc = dh * go[bt] * (1.f - tc * tc) + dcar[bt];
      dcar[bt] = dc * gf[bt];
      bf16x4 dg4;  // i, f, g, o; rows beyond B hold zeros (their activations were read as zeros)
      dg4[0] = (bf16)(dc * gg[bt] * gi[bt] * (1.f - gi[bt]));
      dg4[1] = (bf16)(dc * cp[bt] * gf[bt] * (1.f - gf[bt]));
      dg4[2] = (bf16)(dc * gi[bt] * (1.f - gg[bt] * gg[bt]));
      dg4[3] = (bf16)(dh * tc * go[bt] * (1.f - go[bt]));
      *reinterpret_cast<bf16x4*>(sP + (bt * 16 + l15) * 64 + (4 * wave + q) * 4) = dg4;
    }
    __syncthreads();
    {
      // this workgroup's 32 x 64 values: (a) one 128-byte line per batch row to the exchange buffer, (b) per gate
      // 16 units = 32 bytes to dgates (gate-major), which the sequence-wide GEMMs read after the kernel
      const int row = tid >> 3, c8 = tid & 7;
      if (has_prev && row < bv) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(sP + row * 64 + c8 * 8);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (ts & 1) * parity_bytes + (row * K + 4 * u0 + c8 * 8) * 2, 0, 16);
      }
      const int g4 = c8 >> 1, half = c8 & 1;
      const int b = b0 + row;
      if (b < a.B && row < bv) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sP[row * 64 + (8 * half + e) * 4 + g4];
        *reinterpret_cast<bf16x8*>(D.dgates + ((size_t)t * a.B + b) * K + g4 * H + u0 + 8 * half) = o;
      }
    }
    if (has_prev) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + blockIdx.x, (unsigned)(ts + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __syncthreads();  // sP is rewritten by the next step
    }
  }
}

// Exchange buffer of the entry point that allocates nothing (vc_lstm_persist_fwd_noalloc): it lives in the code
// object, one copy per GPU the library runs on, held until the process ends.  Host code never uses the array's name
// as an address: the per-device address comes from hipGetSymbolAddress (scratch_ptr below).
constexpr size_t SCRATCH_BYTES = 512u << 10;  // e.g. both directions of B <= 64, H = 512 or of B <= 32, H = 1024
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[SCRATCH_BYTES];

constexpr int MAX_DEVICES = VC_MAX_DEVICES;
void* g_scratch_ptr[MAX_DEVICES];  // zero-initialised; written once per device
std::mutex g_scratch_mutex;

// Device address of `scratch` on the current device (looked up once per device, with that device current).
int scratch_ptr(void** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
    vc_set_error("vc_lstm_persist_fwd_noalloc: no usable current device (hipGetDevice gave %d)", dev);
    return VC_ERR_HIP;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
      vc_set_error("vc_lstm_persist_fwd_noalloc: hipGetSymbolAddress failed on device %d: %s", dev,
                   hipGetErrorString(e));
      return VC_ERR_HIP;
    }
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return VC_OK;
}

template <int NKS>
int launch_lstm_bwd(const LstmBwdArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
  static bool attr_set = false;  // > 64 KB of dynamic LDS has to be allowed once per kernel
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_persist_bwd_kernel<NKS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vc_set_error("vc_lstm_persist_bwd: cannot reserve %zu bytes of LDS", lds);
      return VC_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(lstm_persist_bwd_kernel<NKS>, grid, dim3(LTHR), lds, stream, a);
  return VC_OK;
}

}  // namespace

extern "C" {

// Size in bytes of the exchange buffer one direction needs (zeroed by vc_lstm_persist_fwd itself).
long long vc_lstm_persist_xchg_bytes(int B, int H) {
  const long long chunks = (B + BCHUNK - 1) / BCHUNK;
  return chunks * 2 * (H / 2) * BCHUNK * 8;
}

// 1 if (B, H, ndir) can run as one persistent launch: H a power of two in 32..1024 and every workgroup
// resident at once (H/16 x ndir x ceil(B/32) <= the device's CU count).
int vc_lstm_persist_supported(int dtype, int B, int H, int ndir) {
  if (dtype != VC_BF16 || B <= 0 || ndir < 1 || ndir > 2) return 0;
  if (H < 32 || H > 1024 || (H & (H - 1))) return 0;
  const long long wgs = (long long)(H / UNITS) * ndir * ((B + BCHUNK - 1) / BCHUNK);
  return wgs <= vc_device_cus() ? 1 : 0;
}

// All T steps of one LSTM layer, ndir directions (direction 1 runs the sequence in reverse).
// Per direction d: gates_x[d] [T][B][4H] fp32 (x W_ih^T + biases), w_hh[d] [4H][H] bf16 (gate order i,f,g,o),
// hs[d] receives h_t at hs[d][(t*B + b)*ldh + j], act[d] [T][B][4H] and c[d] [T][B][H] fp32 are the values the
// backward pass needs, xchg[d] is scratch of vc_lstm_persist_xchg_bytes(B, H) bytes.  err (nullable) is set to
// 1 if a workgroup gave up waiting for its peers.
int vc_lstm_persist_fwd(int dtype, const float* const* gates_x, const void* const* w_hh, void* const* hs, int ldh,
                        float* const* act, float* const* c, void* const* xchg, int* err, int T, int B, int H,
                        int ndir, hipStream_t stream) {
  VC_CHECK_ARG(gates_x && w_hh && hs && act && c && xchg, "vc_lstm_persist_fwd: null pointer");
  VC_CHECK_SHAPE(vc_lstm_persist_supported(dtype, B, H, ndir) && T > 0 && ldh >= H,
                 "vc_lstm_persist_fwd: unsupported shape B=%d H=%d ndir=%d (ask vc_lstm_persist_supported)", B, H,
                 ndir);
  LstmArgs a;
  a.T = T; a.B = B; a.H = H; a.ldh = ldh; a.err = err;
  const long long xb = vc_lstm_persist_xchg_bytes(B, H);
  for (int d = 0; d < ndir; ++d) {
    VC_CHECK_ARG(gates_x[d] && w_hh[d] && hs[d] && act[d] && c[d] && xchg[d], "vc_lstm_persist_fwd: null pointer");
    a.d[d].gates_x = gates_x[d];
    a.d[d].w_hh = (const bf16*)w_hh[d];
    a.d[d].hs = (bf16*)hs[d];
    a.d[d].act = act[d];
    a.d[d].c = c[d];
    a.d[d].xchg = (unsigned long long*)xchg[d];
    a.d[d].reverse = d;
    // tags must not survive from an earlier launch (a replayed graph repeats this memset node first)
    if (hipMemsetAsync(xchg[d], 0, (size_t)xb, stream) != hipSuccess) {
      vc_set_error("vc_lstm_persist_fwd: hipMemsetAsync failed");
      return VC_ERR_HIP;
    }
  }
  if (ndir == 1) a.d[1] = a.d[0];
  const dim3 grid(H / UNITS, ndir, (B + BCHUNK - 1) / BCHUNK);
  const size_t lds = (size_t)BCHUNK * H * sizeof(bf16);
  switch (H / 32) {
    case 1: hipLaunchKernelGGL(lstm_persist_fwd_kernel<1>, grid, dim3(LTHR), lds, stream, a); break;
    case 2: hipLaunchKernelGGL(lstm_persist_fwd_kernel<2>, grid, dim3(LTHR), lds, stream, a); break;
    case 4: hipLaunchKernelGGL(lstm_persist_fwd_kernel<4>, grid, dim3(LTHR), lds, stream, a); break;
    case 8: hipLaunchKernelGGL(lstm_persist_fwd_kernel<8>, grid, dim3(LTHR), lds, stream, a); break;
    case 16: hipLaunchKernelGGL(lstm_persist_fwd_kernel<16>, grid, dim3(LTHR), lds, stream, a); break;
    default: hipLaunchKernelGGL(lstm_persist_fwd_kernel<32>, grid, dim3(LTHR), lds, stream, a); break;
  }
  VC_LAUNCH_CHECK("vc_lstm_persist_fwd");
  return VC_OK;
}

// vc_lstm_persist_fwd for callers that cannot allocate device memory (code under HIP graph capture, serving loops,
// harnesses that forbid hipMalloc): the exchange buffer is the library's own `scratch` array instead of a caller
// buffer; the call allocates and frees nothing.  Shapes whose exchange needs more than the array holds
// (ndir * ceil(B/32) * H * 256 bytes > 512 KiB) are refused.  Every call on a GPU uses that GPU's one array: calls
// that overlap in time on one GPU (two streams) race; calls on one stream are ordered and each clears the tags
// of the one before.
int vc_lstm_persist_fwd_noalloc(int dtype, const float* const* gates_x, const void* const* w_hh, void* const* hs,
                                int ldh, float* const* act, float* const* c, int* err, int T, int B, int H, int ndir,
                                hipStream_t stream) {
  VC_CHECK_SHAPE(vc_lstm_persist_supported(dtype, B, H, ndir),
                 "vc_lstm_persist_fwd_noalloc: unsupported shape B=%d H=%d ndir=%d (ask vc_lstm_persist_supported)", B,
                 H, ndir);
  const long long per_dir = vc_lstm_persist_xchg_bytes(B, H);
  VC_CHECK_SHAPE(per_dir * ndir <= (long long)SCRATCH_BYTES,
                 "vc_lstm_persist_fwd_noalloc: B=%d H=%d ndir=%d needs %lld bytes of exchange buffer, the built-in one "
                 "holds %zu; use vc_lstm_persist_fwd with a caller buffer", B, H, ndir, per_dir * ndir, SCRATCH_BYTES);
  void* base = nullptr;
  const int rc = scratch_ptr(&base);
  if (rc) return rc;
  void* xchg[2] = {base, static_cast<unsigned char*>(base) + per_dir};
  return vc_lstm_persist_fwd(dtype, gates_x, w_hh, hs, ldh, act, c, xchg, err, T, B, H, ndir, stream);
}

// ---- backward ----
long long vc_lstm_persist_bwd_xchg_bytes(int B, int H) {
  const long long chunks = (B + BCHUNK - 1) / BCHUNK;
  return chunks * (2LL * BCHUNK * 4 * H * (long long)sizeof(bf16) + BWD_FLAG_WORDS * 4);
}

// 1 if the backward recurrence of (B, H, ndir) can run as one persistent launch: H a power of two in 32..512
// (the gate gradients of 32 batch rows, 32 x 4H bf16, live in one workgroup's LDS) and every workgroup resident
// at once, one per CU.
int vc_lstm_persist_bwd_supported(int dtype, int B, int H, int ndir) {
  if (dtype != VC_BF16 || B <= 0 || ndir < 1 || ndir > 2) return 0;
  if (H < 32 || H > 512 || (H & (H - 1))) return 0;
  const long long wgs = (long long)(H / UNITS) * ndir * ((B + BCHUNK - 1) / BCHUNK);
  return wgs <= vc_device_cus() ? 1 : 0;
}

// Backward recurrence of one LSTM layer over all T steps, ndir directions in one launch.  Per direction d:
// dh_out[d] is the gradient of the layer output at dh_out[d][(t*B + b)*ldo + j] (bf16), act[d] / c[d] what
// vc_lstm_persist_fwd (or vc_lstm_cell_fwd) saved, w_hh_t[d] = W_hh^T [H][4H] bf16, dgates[d] [T][B][4H] bf16
// receives the gate pre-activation gradients of every step (gate order i,f,g,o), xchg[d] is scratch of
// vc_lstm_persist_bwd_xchg_bytes(B, H) bytes (zeroed by the call).  Direction 1 walks the sequence forward in
// time (it ran backward in the forward pass).  *err as for vc_lstm_persist_fwd.
int vc_lstm_persist_bwd(int dtype, const void* const* dh_out, int ldo, const float* const* act, const float* const* c,
                        const void* const* w_hh_t, void* const* dgates, void* const* xchg, int* err, int T, int B,
                        int H, int ndir, hipStream_t stream) {
  VC_CHECK_ARG(dh_out && act && c && w_hh_t && dgates && xchg, "vc_lstm_persist_bwd: null pointer");
  VC_CHECK_SHAPE(vc_lstm_persist_bwd_supported(dtype, B, H, ndir) && T > 0 && ldo >= H,
                 "vc_lstm_persist_bwd: unsupported shape B=%d H=%d ndir=%d (ask vc_lstm_persist_bwd_supported)", B, H,
                 ndir);
  LstmBwdArgs a;
  a.T = T; a.B = B; a.H = H; a.ldo = ldo; a.err = err;
  a.bvs = B > 16 ? 5 : (B > 8 ? 4 : 3);
  const long long xb = vc_lstm_persist_bwd_xchg_bytes(B, H);
  for (int d = 0; d < ndir; ++d) {
    VC_CHECK_ARG(dh_out[d] && act[d] && c[d] && w_hh_t[d] && dgates[d] && xchg[d], "vc_lstm_persist_bwd: null pointer");
    a.d[d].dh_out = (const bf16*)dh_out[d];
    a.d[d].act = act[d];
    a.d[d].c = c[d];
    a.d[d].w_hh_t = (const bf16*)w_hh_t[d];
    a.d[d].dgates = (bf16*)dgates[d];
    a.d[d].xchg = (unsigned char*)xchg[d];
    a.d[d].reverse = d;
    if (hipMemsetAsync(xchg[d], 0, (size_t)xb, stream) != hipSuccess) {
      vc_set_error("vc_lstm_persist_bwd: hipMemsetAsync failed");
      return VC_ERR_HIP;
    }
  }
  if (ndir == 1) a.d[1] = a.d[0];
  const dim3 grid(H / UNITS, ndir, (B + BCHUNK - 1) / BCHUNK);
  size_t lds = (size_t)BCHUNK * 4 * H * sizeof(bf16) + 4 * 16 * 32 * sizeof(float) + BCHUNK * 64 * sizeof(bf16);
  if (lds < 84 * 1024) lds = 84 * 1024;  // one workgroup per CU, the configuration the flag hand-off was validated in
  int rc;
  switch (H / 32) {
    case 1: rc = launch_lstm_bwd<1>(a, grid, lds, stream); break;
    case 2: rc = launch_lstm_bwd<2>(a, grid, lds, stream); break;
    case 4: rc = launch_lstm_bwd<4>(a, grid, lds, stream); break;
    case 8: rc = launch_lstm_bwd<8>(a, grid, lds, stream); break;
    default: rc = launch_lstm_bwd<16>(a, grid, lds, stream); break;
  }
  if (rc) return rc;
  VC_LAUNCH_CHECK("vc_lstm_persist_bwd");
  return VC_OK;
}

}  // extern "C"
