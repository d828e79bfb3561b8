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

}  // extern "C"
