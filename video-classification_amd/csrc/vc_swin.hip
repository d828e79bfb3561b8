// Shifted-window attention and patch merging for the Swin-T backbone (gfx950).
//
// Window partition, the cyclic shift and its inverse are never materialised: the kernels address the
// token matrix [B][H][W][*] through the window/shift index map.  A (window, head) unit is 49 x 49 x 32
// (window 7, head_dim 32): far too small to be worth an MFMA pipeline next to the QKV / projection / MLP
// GEMMs (the attention products are ~3 % of Swin-T's FLOPs), so one wave owns one unit, K/V/Q live in LDS,
// each lane owns one query row and the softmax never leaves registers.
//
// qkv layout (timm): row = token, column = which*C + head*32 + e, which in {q, k, v}.
#include "vc_common.h"

namespace {

constexpr int HD = 32;      // head dim
constexpr int MAXN = 49;    // tokens per window (window 7)
constexpr int LDP = 50;     // padded row length of the [i][j] LDS matrices

struct SwinGeom {
  int B, H, W, C, heads, ws, shift;
  int nwh, nww;  // windows per column / row
};

// token index (row of the [B*H*W] matrices) of local position n in window (b, wh, ww)
__device__ __forceinline__ int token_of(const SwinGeom& g, int b, int wh, int ww, int n, int& ph, int& pw) {
  const int i = n / g.ws, j = n - i * g.ws;
  ph = wh * g.ws + i;  // coordinates in the rolled frame
  pw = ww * g.ws + j;
  int h = ph + g.shift, w = pw + g.shift;
  if (h >= g.H) h -= g.H;
  if (w >= g.W) w -= g.W;
  return (b * g.H + h) * g.W + w;
}

__device__ __forceinline__ int region_of(int p, int L, int ws, int shift) {
  return p < L - ws ? 0 : (p < L - shift ? 1 : 2);
}

template <typename T>
__device__ __forceinline__ void load_row32(const T* p, float* dst) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const F8 v = load8(p + 8 * k);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[8 * k + e] = v.v[e];
  }
}

// scores of query row `lane` against all keys: p[j] = softmax_j(scale*q.k_j + bias + mask); returns nothing,
// p[] holds probabilities on exit.  sK is [N][HD] fp32 in LDS.
__device__ __forceinline__ void scores_softmax(const SwinGeom& g, const float* q, const float* sK,
                                               const float* __restrict__ bias_table, int head, int lane, int N,
                                               int my_reg, const int* sReg, float scale, float* p) {
  const int ws = g.ws;
  const int qi = lane / ws, qj = lane - qi * ws;
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXN; ++j) {
    float s = 0.f;
    if (j < N) {
      const float* kr = sK + j * HD;
#pragma unroll
      for (int e = 0; e < HD; e += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + e);
        s = fmaf(q[e], kv[0], s);
        s = fmaf(q[e + 1], kv[1], s);
        s = fmaf(q[e + 2], kv[2], s);
        s = fmaf(q[e + 3], kv[3], s);
      }
      const int kj_i = j / ws, kj_j = j - kj_i * ws;
      const int idx = (qi - kj_i + ws - 1) * (2 * ws - 1) + (qj - kj_j + ws - 1);
      s = s * scale + bias_table[idx * g.heads + head];
      if (g.shift && sReg[j] != my_reg) s -= 100.f;
      mx = fmaxf(mx, s);
    } else {
      s = -INFINITY;
    }
    p[j] = s;
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < MAXN; ++j) {
    const float e = (j < N) ? __expf(p[j] - mx) : 0.f;
    p[j] = e;
    sum += e;
  }
  const float inv = 1.f / sum;
#pragma unroll
  for (int j = 0; j < MAXN; ++j) p[j] *= inv;
}

// one wave per (window, head)
template <typename T>
__global__ __launch_bounds__(64) void window_attn_fwd_kernel(SwinGeom g, const T* __restrict__ qkv,
                                                             const float* __restrict__ bias_table, T* __restrict__ out,
                                                             float scale) {
  __shared__ __attribute__((aligned(16))) float sK[MAXN * HD];
  __shared__ __attribute__((aligned(16))) float sV[MAXN * HD];
  __shared__ int sReg[64];
  const int lane = threadIdx.x;
  const int N = g.ws * g.ws;
  int unit = blockIdx.x;
  const int head = unit % g.heads;
  unit /= g.heads;
  const int ww = unit % g.nww;
  unit /= g.nww;
  const int wh = unit % g.nwh;
  const int b = unit / g.nwh;

  float q[HD];
  int tok = 0, my_reg = 0;
  if (lane < N) {
    int ph, pw;
    tok = token_of(g, b, wh, ww, lane, ph, pw);
    my_reg = region_of(ph, g.H, g.ws, g.shift) * 3 + region_of(pw, g.W, g.ws, g.shift);
    const T* row = qkv + (size_t)tok * 3 * g.C + head * HD;
    load_row32(row, q);
    load_row32(row + g.C, sK + lane * HD);
    load_row32(row + 2 * g.C, sV + lane * HD);
  } else {
#pragma unroll
    for (int e = 0; e < HD; ++e) q[e] = 0.f;
  }
  sReg[lane] = my_reg;
  __syncthreads();

  float p[MAXN];
  scores_softmax(g, q, sK, bias_table, head, lane < N ? lane : 0, N, my_reg, sReg, scale, p);
  float o[HD];
#pragma unroll
  for (int e = 0; e < HD; ++e) o[e] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXN; ++j) {
    if (j < N) {
      const float* vr = sV + j * HD;
#pragma unroll
      for (int e = 0; e < HD; e += 4) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + e);
        o[e] = fmaf(p[j], vv[0], o[e]);
        o[e + 1] = fmaf(p[j], vv[1], o[e + 1]);
        o[e + 2] = fmaf(p[j], vv[2], o[e + 2]);
        o[e + 3] = fmaf(p[j], vv[3], o[e + 3]);
      }
    }
  }
  if (lane < N) {
    T* orow = out + (size_t)tok * g.C + head * HD;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      F8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v.v[e] = o[8 * k + e];
      store8(orow + 8 * k, v);
    }
  }
}

// One wave per (head, chunk of windows): recomputes the probabilities, produces dqkv for its windows and
// accumulates the relative-position-bias gradient of its windows in registers (lane i owns row i of dS).
template <typename T>
__global__ __launch_bounds__(64) void window_attn_bwd_kernel(SwinGeom g, const T* __restrict__ qkv,
                                                             const T* __restrict__ dout,
                                                             const float* __restrict__ bias_table,
                                                             T* __restrict__ dqkv, float* __restrict__ dbias_table,
                                                             float scale, int windows_total, int windows_per_block) {
  __shared__ __attribute__((aligned(16))) float sQ[MAXN * HD];
  __shared__ __attribute__((aligned(16))) float sK[MAXN * HD];
  __shared__ __attribute__((aligned(16))) float sV[MAXN * HD];
  __shared__ __attribute__((aligned(16))) float sDO[MAXN * HD];
  __shared__ float sP[MAXN * LDP];
  __shared__ float sDS[MAXN * LDP];
  __shared__ int sReg[64];
  const int lane = threadIdx.x;
  const int N = g.ws * g.ws;
  const int head = blockIdx.y;
  const int w_begin = blockIdx.x * windows_per_block;
  const int w_end = min(windows_total, w_begin + windows_per_block);

  float db[MAXN];
#pragma unroll
  for (int j = 0; j < MAXN; ++j) db[j] = 0.f;

  for (int win = w_begin; win < w_end; ++win) {
    int u = win;
    const int ww = u % g.nww;
    u /= g.nww;
    const int wh = u % g.nwh;
    const int b = u / g.nwh;
    float q[HD], dO[HD];
    int tok = 0, my_reg = 0;
    __syncthreads();  // previous window's LDS reads are finished
    if (lane < N) {
      int ph, pw;
      tok = token_of(g, b, wh, ww, lane, ph, pw);
      my_reg = region_of(ph, g.H, g.ws, g.shift) * 3 + region_of(pw, g.W, g.ws, g.shift);
      const T* row = qkv + (size_t)tok * 3 * g.C + head * HD;
      load_row32(row, q);
      load_row32(row + g.C, sK + lane * HD);
      load_row32(row + 2 * g.C, sV + lane * HD);
      load_row32(dout + (size_t)tok * g.C + head * HD, dO);
#pragma unroll
      for (int e = 0; e < HD; ++e) {
        sQ[lane * HD + e] = q[e];
        sDO[lane * HD + e] = dO[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < HD; ++e) q[e] = dO[e] = 0.f;
    }
    sReg[lane] = my_reg;
    __syncthreads();

    float p[MAXN];
    scores_softmax(g, q, sK, bias_table, head, lane < N ? lane : 0, N, my_reg, sReg, scale, p);
    // dP[i][j] = dO_i . V_j ;  dS = P * (dP - sum_j P dP)
    float ds[MAXN];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      float d = 0.f;
      if (j < N) {
        const float* vr = sV + j * HD;
#pragma unroll
        for (int e = 0; e < HD; e += 4) {
          const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + e);
          d = fmaf(dO[e], vv[0], d);
          d = fmaf(dO[e + 1], vv[1], d);
          d = fmaf(dO[e + 2], vv[2], d);
          d = fmaf(dO[e + 3], vv[3], d);
        }
      }
      ds[j] = d;
      dot = fmaf(p[j], d, dot);
    }
    float dq[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) dq[e] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      const float s = (j < N && lane < N) ? p[j] * (ds[j] - dot) : 0.f;
      ds[j] = s;
      db[j] += s;
      if (j < N) {
        const float* kr = sK + j * HD;
#pragma unroll
        for (int e = 0; e < HD; e += 4) {
          const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + e);
          dq[e] = fmaf(s, kv[0], dq[e]);
          dq[e + 1] = fmaf(s, kv[1], dq[e + 1]);
          dq[e + 2] = fmaf(s, kv[2], dq[e + 2]);
          dq[e + 3] = fmaf(s, kv[3], dq[e + 3]);
        }
      }
    }
    if (lane < N) {
#pragma unroll
      for (int j = 0; j < MAXN; ++j) {
        if (j < N) {
          sP[lane * LDP + j] = p[j];
          sDS[lane * LDP + j] = ds[j];
        }
      }
    }
    __syncthreads();
    // lane j: dK_j = scale * sum_i dS[i][j] Q_i ;  dV_j = sum_i P[i][j] dO_i
    float dk[HD], dv[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) dk[e] = dv[e] = 0.f;
    if (lane < N) {
      for (int i = 0; i < N; ++i) {
        const float s = sDS[i * LDP + lane], pp = sP[i * LDP + lane];
        const float* qr = sQ + i * HD;
        const float* dr = sDO + i * HD;
#pragma unroll
        for (int e = 0; e < HD; e += 4) {
          const f32x4 qv = *reinterpret_cast<const f32x4*>(qr + e);
          const f32x4 dvv = *reinterpret_cast<const f32x4*>(dr + e);
          dk[e] = fmaf(s, qv[0], dk[e]);
          dk[e + 1] = fmaf(s, qv[1], dk[e + 1]);
          dk[e + 2] = fmaf(s, qv[2], dk[e + 2]);
          dk[e + 3] = fmaf(s, qv[3], dk[e + 3]);
          dv[e] = fmaf(pp, dvv[0], dv[e]);
          dv[e + 1] = fmaf(pp, dvv[1], dv[e + 1]);
          dv[e + 2] = fmaf(pp, dvv[2], dv[e + 2]);
          dv[e + 3] = fmaf(pp, dvv[3], dv[e + 3]);
        }
      }
      T* drow = dqkv + (size_t)tok * 3 * g.C + head * HD;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        F8 a, bq, c;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          a.v[e] = dq[8 * k + e] * scale;
          bq.v[e] = dk[8 * k + e] * scale;
          c.v[e] = dv[8 * k + e];
        }
        store8(drow + 8 * k, a);
        store8(drow + g.C + 8 * k, bq);
        store8(drow + 2 * g.C + 8 * k, c);
      }
    }
  }
  if (lane < N && dbias_table) {
    const int ws = g.ws;
    const int qi = lane / ws, qj = lane - qi * ws;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      if (j < N) {
        const int ki = j / ws, kj = j - ki * ws;
        const int idx = (qi - ki + ws - 1) * (2 * ws - 1) + (qj - kj + ws - 1);
        atomicAdd(dbias_table + idx * g.heads + head, db[j]);
      }
    }
  }
}

// PatchMerging gather: [B][H][W][C] -> [B][H/2][W/2][4C], channel blocks ordered (h even,w even), (h odd,w even),
// (h even,w odd), (h odd,w odd).  inverse = 1 scatters a gradient back.
template <typename T>
__global__ void patch_merge_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C8,
                                   int inverse) {
  const int H2 = H / 2, W2 = W / 2;
  const int64_t total = (int64_t)B * H2 * W2 * 4 * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % C8);
    int64_t t = i / C8;
    const int blk = (int)(t % 4);
    t /= 4;
    const int w2 = (int)(t % W2);
    t /= W2;
    const int h2 = (int)(t % H2);
    const int b = (int)(t / H2);
    const int h = 2 * h2 + (blk & 1), w = 2 * w2 + (blk >> 1);
    const size_t full = ((((size_t)b * H + h) * W + w) * C8 + cc) * 8;
    if (!inverse) store8(out + i * 8, load8(in + full));
    else store8(out + full, load8(in + i * 8));
  }
}

int fill_swin(SwinGeom& g, int B, int H, int W, int C, int heads, int ws, int shift, const char* who) {
  VC_CHECK_SHAPE(B > 0 && H > 0 && W > 0 && ws > 0 && ws * ws <= MAXN && H % ws == 0 && W % ws == 0,
                 "%s: window %d does not tile %dx%d (or exceeds 7x7)", who, ws, H, W);
  VC_CHECK_SHAPE(heads > 0 && C == heads * HD, "%s: head_dim must be 32 (C=%d, heads=%d)", who, C, heads);
  VC_CHECK_SHAPE(shift >= 0 && shift < ws, "%s: bad shift %d", who, shift);
  g.B = B; g.H = H; g.W = W; g.C = C; g.heads = heads; g.ws = ws; g.shift = shift;
  g.nwh = H / ws; g.nww = W / ws;
  return VC_OK;
}

}  // namespace

extern "C" {

int vc_window_attn_fwd(int dtype, const void* qkv, const float* bias_table, void* out, int B, int H, int W, int C,
                       int heads, int ws, int shift, float scale, hipStream_t stream) {
  VC_CHECK_ARG(qkv && bias_table && out, "vc_window_attn_fwd: null pointer");
  SwinGeom g;
  int rc = fill_swin(g, B, H, W, C, heads, ws, shift, "vc_window_attn_fwd");
  if (rc) return rc;
  const int units = B * g.nwh * g.nww * heads;
  if (dtype == VC_BF16)
    hipLaunchKernelGGL((window_attn_fwd_kernel<bf16>), dim3(units), dim3(64), 0, stream, g, (const bf16*)qkv,
                       bias_table, (bf16*)out, scale);
  else if (dtype == VC_F32)
    hipLaunchKernelGGL((window_attn_fwd_kernel<float>), dim3(units), dim3(64), 0, stream, g, (const float*)qkv,
                       bias_table, (float*)out, scale);
  else { vc_set_error("vc_window_attn_fwd: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_window_attn_fwd");
  return VC_OK;
}

// dbias_table ([(2ws-1)^2][heads] fp32) is accumulated into with atomics
int vc_window_attn_bwd(int dtype, const void* qkv, const void* dout, const float* bias_table, void* dqkv,
                       float* dbias_table, int B, int H, int W, int C, int heads, int ws, int shift, float scale,
                       hipStream_t stream) {
  VC_CHECK_ARG(qkv && dout && bias_table && dqkv, "vc_window_attn_bwd: null pointer");
  SwinGeom g;
  int rc = fill_swin(g, B, H, W, C, heads, ws, shift, "vc_window_attn_bwd");
  if (rc) return rc;
  const int windows = B * g.nwh * g.nww;
  int wpb = 1;
  while ((int64_t)ceil_div(windows, wpb) * heads > 8192) wpb *= 2;
  dim3 grid(ceil_div(windows, wpb), heads);
  if (dtype == VC_BF16)
    hipLaunchKernelGGL((window_attn_bwd_kernel<bf16>), grid, dim3(64), 0, stream, g, (const bf16*)qkv,
                       (const bf16*)dout, bias_table, (bf16*)dqkv, dbias_table, scale, windows, wpb);
  else if (dtype == VC_F32)
    hipLaunchKernelGGL((window_attn_bwd_kernel<float>), grid, dim3(64), 0, stream, g, (const float*)qkv,
                       (const float*)dout, bias_table, (float*)dqkv, dbias_table, scale, windows, wpb);
  else { vc_set_error("vc_window_attn_bwd: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_window_attn_bwd");
  return VC_OK;
}

int vc_patch_merge(int dtype, const void* in, void* out, int B, int H, int W, int C, int inverse,
                   hipStream_t stream) {
  VC_CHECK_ARG(in && out, "vc_patch_merge: null pointer");
  VC_CHECK_SHAPE(B > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "vc_patch_merge: need even H, W and C %% 8 == 0");
  const int64_t total = (int64_t)B * H * W * (C / 8);
  int64_t blocks = ceil_div64(total, 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == VC_BF16)
    hipLaunchKernelGGL((patch_merge_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)in,
                       (bf16*)out, B, H, W, C / 8, inverse);
  else if (dtype == VC_F32)
    hipLaunchKernelGGL((patch_merge_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)in,
                       (float*)out, B, H, W, C / 8, inverse);
  else { vc_set_error("vc_patch_merge: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_patch_merge");
  return VC_OK;
}

}  // extern "C"
