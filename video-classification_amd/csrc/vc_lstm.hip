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
// for its 16 units j, i.e. the COMPLETE gate-gradient vector of the step processed before (4x the data the
// forward pass exchanges).  The owner of (b, unit) computes the unit's four gate gradients in registers from
// dh = dh_out + dh_rec, the carried dc and the saved activations, stores them to dgates (the sequence-wide
// weight / input gradient GEMMs read them afterwards) and publishes them as tagged granules.  The reduction
// index is ordered k = gate*H + unit, the order of a row of W_hh^T, so a wave's A fragments are 16-byte loads
// of W_hh^T [H][4H]; the four waves split k and their 16x32 partial products are summed through LDS.
// ---------------------------------------------------------------------------------------------------------
struct LstmBwdDir {
  const bf16* dh_out;   // gradient of the layer output: dh_out[(t*B + b)*ldo + unit]
  const float* act;     // [T][B][4H] post-activation gates saved by the forward pass
  const float* c;       // [T][B][H]
  const bf16* w_hh_t;   // [H][4H]: W_hh^T
  bf16* dgates;         // [T][B][4H], gate-major like act
  unsigned long long* xchg;  // [chunks][2 parities][2H * 32] granules
  int reverse;
};
struct LstmBwdArgs {
  LstmBwdDir d[2];
  int T, B, H, ldo;
  int bvs;  // log2 of the batch rows a chunk exchanges (8, 16 or 32: rows beyond B in the last chunk are zeros)
  int* err;
};

template <int NKS>  // NKS = H / 32 = k-steps of the recurrent product per wave (4H / 32 in all)
__global__ __launch_bounds__(LTHR, 1) void lstm_persist_bwd_kernel(LstmBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int H = NKS * 32, K = 4 * H;
  bf16* sD = reinterpret_cast<bf16*>(lds_raw);                    // [32][K] gate gradients of the previous step
  float* sR = reinterpret_cast<float*>(lds_raw + (size_t)BCHUNK * K * sizeof(bf16));  // [4][16][32] partials
  const LstmBwdDir D = blockIdx.y ? a.d[1] : a.d[0];  // a runtime index would put the struct in scratch
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int u0 = blockIdx.x * UNITS;
  const int b0 = blockIdx.z * BCHUNK;
  const int unit = u0 + 4 * wave + q;
  const int bv = 1 << a.bvs;
  const int ntile = bv > 16 ? 2 : 1;  // batch tiles of 16 that carry data

  // this wave's slice of W_hh^T: rows = the workgroup's 16 units, k in [wave*H, (wave+1)*H)
  bf16x8 wreg[NKS];
  {
    const bf16* wrow = D.w_hh_t + (size_t)(u0 + l15) * K + wave * H + 8 * q;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) wreg[ks] = *reinterpret_cast<const bf16x8*>(wrow + 32 * ks);
  }
  // rows that are never exchanged stay zero for the whole kernel
  for (int i = tid; i < BCHUNK * K / 8; i += LTHR) reinterpret_cast<bf16x8*>(sD)[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();

  gu64* xbase = (gu64*)(D.xchg + (size_t)blockIdx.z * 2 * (K / 2) * BCHUNK);
  const int per_parity = (K / 2) * BCHUNK;
  const int ng = (K / 2 * bv) / LTHR;  // granules per thread per step (>= 1: K/2*bv >= 64*8*... see host check)

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
      gu64* xp = xbase + ((ts - 1) & 1) * per_parity;
      for (int base = 0; base < ng; base += 32) {
        unsigned long long g[32];
        unsigned spins = 0;
        for (;;) {
#pragma unroll
          for (int k = 0; k < 32; ++k)
            if (base + k < ng) g[k] = __hip_atomic_load(xp + tid + LTHR * (base + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bool ok = true;
#pragma unroll
          for (int k = 0; k < 32; ++k)
            if (base + k < ng) ok &= (unsigned)(g[k] >> 32) == (unsigned)ts;
          if (ok) break;
          if (++spins > SPIN_LIMIT) {
            if (a.err) atomicExch(a.err, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          if (base + k < ng) {
            const int idx = tid + LTHR * (base + k);
            const int kp = idx >> a.bvs, b = idx & (bv - 1);  // k pair, batch row
            const int kk = 2 * kp;
            const int chunk = (kk >> 3) ^ (b & 15);
            *reinterpret_cast<unsigned*>(sD + b * K + chunk * 8 + (kk & 7)) = (unsigned)g[k];
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
    float dgv[2][4];
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const int b = b0 + bt * 16 + l15;
      const float dh = dho[bt] + dhr[bt];
      const float tc = tanhf(cc[bt]);
      const float dc = dh * go[bt] * (1.f - tc * tc) + dcar[bt];
      dcar[bt] = dc * gf[bt];
      dgv[bt][0] = dc * gg[bt] * gi[bt] * (1.f - gi[bt]);
      dgv[bt][1] = dc * cp[bt] * gf[bt] * (1.f - gf[bt]);
      dgv[bt][2] = dc * gi[bt] * (1.f - gg[bt] * gg[bt]);
      dgv[bt][3] = dh * tc * go[bt] * (1.f - go[bt]);
      if (b < a.B && bt < ntile) {
        bf16* dp = D.dgates + ((size_t)t * a.B + b) * K + unit;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) dp[g4 * H] = (bf16)dgv[bt][g4];
      } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) dgv[bt][g4] = 0.f;
      }
    }
    if (ts + 1 < a.T) {
      // publish: a granule holds one gate of two adjacent units (even unit low); the even-unit lane sends
      // gates i and f, its odd neighbour (lane ^ 16) sends g and o
      gu64* xp = xbase + (ts & 1) * per_parity;
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
        float other[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) other[g4] = __shfl_xor(dgv[bt][g4], 16, 64);
        const int bl = bt * 16 + l15;
        if (bl < bv) {
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int g4 = (q & 1) * 2 + h2;
            union {
              bf16 h[2];
              unsigned u;
            } pk;
            // static register indices only (a lane-dependent index would send the arrays to scratch)
            pk.h[0] = (bf16)((q & 1) ? other[2 + h2] : dgv[bt][h2]);
            pk.h[1] = (bf16)((q & 1) ? dgv[bt][2 + h2] : other[h2]);
            const int kp = (g4 * H + (unit & ~1)) >> 1;
            const unsigned long long gr = ((unsigned long long)(unsigned)(ts + 1) << 32) | pk.u;
            __hip_atomic_store(xp + (kp << a.bvs) + bl, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    // the next step's gather overwrites sD and its partials overwrite sR: both are behind barriers every wave
    // passes after its last read (the one before the partial sums, and the one after the next gather)
  }
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
// resident at once (H/16 x ndir x ceil(B/32) <= 256 CUs).
int vc_lstm_persist_supported(int dtype, int B, int H, int ndir) {
  if (dtype != VC_BF16 || B <= 0 || ndir < 1 || ndir > 2) return 0;
  if (H < 32 || H > 1024 || (H & (H - 1))) return 0;
  const long long wgs = (long long)(H / UNITS) * ndir * ((B + BCHUNK - 1) / BCHUNK);
  return wgs <= 256 ? 1 : 0;
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


// ---- backward ----
long long vc_lstm_persist_bwd_xchg_bytes(int B, int H) {
  const long long chunks = (B + BCHUNK - 1) / BCHUNK;
  return chunks * 2 * (2LL * H) * BCHUNK * 8;
}

// 1 if the backward recurrence of (B, H, ndir) can run as one persistent launch: H a power of two in 32..512
// (the gate gradients of 32 batch rows, 32 x 4H bf16, live in one workgroup's LDS) and every workgroup resident
// at once, one per CU.
int vc_lstm_persist_bwd_supported(int dtype, int B, int H, int ndir) {
  if (dtype != VC_BF16 || B <= 0 || ndir < 1 || ndir > 2) return 0;
  if (H < 32 || H > 512 || (H & (H - 1))) return 0;
  const long long wgs = (long long)(H / UNITS) * ndir * ((B + BCHUNK - 1) / BCHUNK);
  return wgs <= 256 ? 1 : 0;
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
    a.d[d].xchg = (unsigned long long*)xchg[d];
    a.d[d].reverse = d;
    if (hipMemsetAsync(xchg[d], 0, (size_t)xb, stream) != hipSuccess) {
      vc_set_error("vc_lstm_persist_bwd: hipMemsetAsync failed");
      return VC_ERR_HIP;
    }
  }
  if (ndir == 1) a.d[1] = a.d[0];
  const dim3 grid(H / UNITS, ndir, (B + BCHUNK - 1) / BCHUNK);
  const size_t lds = (size_t)BCHUNK * 4 * H * sizeof(bf16) + 4 * 16 * 32 * sizeof(float);
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
