// Channel + spatial attention gate of the image / density fusion CNN (reference
// models/swintransformer_RNN/image_density_fusion.py:14-87) for channels-last activations x [N][P][C], P = H*W.
//
//   gc[n][c] = sigmoid(W2 relu(W1 avg_p x) + W2 relu(W1 max_p x))          channel gate, W1 [C/r][C], W2 [C][C/r]
//   gs[n][p] = sigmoid(conv7x7([mean_c x*gc, max_c x*gc]))                 spatial gate, Ws [1][2][k][k]
//   out      = relu(x * gc * gs)
//
// Both gates lie in (0, 1), so the ReLU decision is x > 0.  Maxima send their gradient to the FIRST maximal element
// (nn.AdaptiveMaxPool2d, torch.max(dim)).  The passes over x are HBM-bound (forward 3 reads + 1 write, backward
// 4 reads + 2 writes of an x-sized tensor); everything else is per-frame or per-pixel scalars in fp32.
#include "vc_common.h"

namespace {

constexpr int NTHR = 256;

// ---- per-(frame, channel) mean and max over pixels, in two steps ----
// grid (chunks, N): partial sum / max / first arg-max of a chunk of rows
template <typename T>
__global__ __launch_bounds__(NTHR) void chan_pool_partial_kernel(const T* __restrict__ x, float* __restrict__ psum,
                                                                 float* __restrict__ pmax, int* __restrict__ parg,
                                                                 int P, int C, int rows_per_chunk) {
  __shared__ float s_sum[NTHR][8], s_max[NTHR][8];
  __shared__ int s_arg[NTHR][8];
  const int C8 = C >> 3, RL = NTHR / C8;
  const int cl = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int n = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
  const int r0 = chunk * rows_per_chunk, r1 = min(P, r0 + rows_per_chunk);
  float sum[8], mx[8];
  int arg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sum[i] = 0.f;
    mx[i] = -INFINITY;
    arg[i] = 0x7fffffff;
  }
  if (rl < RL)
    for (int r = r0 + rl; r < r1; r += RL) {
      const F8 v = load8(x + ((size_t)n * P + r) * C + cl * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sum[i] += v.v[i];
        if (v.v[i] > mx[i]) {  // rows ascend within a thread: strict > keeps the first
          mx[i] = v.v[i];
          arg[i] = r;
        }
      }
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s_sum[threadIdx.x][i] = sum[i];
    s_max[threadIdx.x][i] = mx[i];
    s_arg[threadIdx.x][i] = arg[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NTHR) {
    const int l = c >> 3, i = c & 7;
    float a = 0.f, m = -INFINITY;
    int g = 0x7fffffff;
    for (int k = 0; k < RL; ++k) {
      const int t = k * C8 + l;
      a += s_sum[t][i];
      if (s_max[t][i] > m || (s_max[t][i] == m && s_arg[t][i] < g)) {
        m = s_max[t][i];
        g = s_arg[t][i];
      }
    }
    const size_t o = ((size_t)n * chunks + chunk) * C + c;
    psum[o] = a;
    pmax[o] = m;
    parg[o] = g;
  }
}

// grid N: folds the chunks (ascending, so ties keep the first row) and evaluates the channel gate
__global__ __launch_bounds__(NTHR) void chan_gate_kernel(const float* __restrict__ psum, const float* __restrict__ pmax,
                                                         const int* __restrict__ parg, const float* __restrict__ W1,
                                                         const float* __restrict__ W2, float* __restrict__ avg,
                                                         float* __restrict__ mx, int* __restrict__ argp,
                                                         float* __restrict__ hid, float* __restrict__ gate, int chunks,
                                                         int P, int C, int Ch) {
  __shared__ float s_avg[512], s_mx[512], s_h[2][32];
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += NTHR) {
    float a = 0.f, m = -INFINITY;
    int g = 0x7fffffff;
    for (int k = 0; k < chunks; ++k) {
      const size_t o = ((size_t)n * chunks + k) * C + c;
      a += psum[o];
      if (pmax[o] > m) {
        m = pmax[o];
        g = parg[o];
      }
    }
    a /= (float)P;
    s_avg[c] = a;
    s_mx[c] = m;
    avg[(size_t)n * C + c] = a;
    mx[(size_t)n * C + c] = m;
    argp[(size_t)n * C + c] = g;
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * Ch) {  // hidden units of the avg path (0..Ch-1) and of the max path
    const int which = threadIdx.x / Ch, j = threadIdx.x % Ch;
    const float* v = which ? s_mx : s_avg;
    float h = 0.f;
    for (int c = 0; c < C; ++c) h = fmaf(W1[j * C + c], v[c], h);
    h = fmaxf(h, 0.f);
    s_h[which][j] = h;
    hid[((size_t)n * 2 + which) * Ch + j] = h;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NTHR) {
    float z = 0.f;
    for (int j = 0; j < Ch; ++j) z = fmaf(W2[c * Ch + j], s_h[0][j] + s_h[1][j], z);
    gate[(size_t)n * C + c] = 1.f / (1.f + __expf(-z));
  }
}

// ---- per-pixel mean / max over channels of x * gc; a pixel is C/8 adjacent lanes (a power of two <= 64) ----
template <typename T>
__global__ __launch_bounds__(NTHR) void spatial_pool_kernel(const T* __restrict__ x, const float* __restrict__ gc,
                                                            float* __restrict__ pooled, int* __restrict__ argc,
                                                            int64_t NP, int P, int C) {
  const int C8 = C >> 3, PPB = NTHR / C8;
  const int cl = threadIdx.x % C8, pl = threadIdx.x / C8;
  for (int64_t q0 = (int64_t)blockIdx.x * PPB; q0 < NP; q0 += (int64_t)gridDim.x * PPB) {
    const int64_t q = q0 + pl;
    float s = 0.f, m = -INFINITY;
    int a = 0x7fffffff;
    if (q < NP) {
      const int n = (int)(q / P);
      const F8 v = load8(x + q * C + cl * 8), g = load8(gc + (size_t)n * C + cl * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float t = v.v[i] * g.v[i];
        s += t;
        if (t > m) {
          m = t;
          a = cl * 8 + i;
        }
      }
    }
    for (int o = 1; o < C8; o <<= 1) {
      s += __shfl_xor(s, o, 64);
      const float m2 = __shfl_xor(m, o, 64);
      const int a2 = __shfl_xor(a, o, 64);
      if (m2 > m || (m2 == m && a2 < a)) {
        m = m2;
        a = a2;
      }
    }
    if (q < NP && cl == 0) {
      pooled[q * 2] = s / (float)C;
      pooled[q * 2 + 1] = m;
      argc[q] = a;
    }
  }
}

// gs = sigmoid(conv k x k over the two pooled planes); one thread per pixel
__global__ __launch_bounds__(NTHR) void spatial_gate_kernel(const float* __restrict__ pooled, const float* __restrict__ Ws,
                                                            float* __restrict__ gs, int N, int H, int W, int k) {
  __shared__ float s_w[2 * 49];
  for (int i = threadIdx.x; i < 2 * k * k; i += NTHR) s_w[i] = Ws[i];
  __syncthreads();
  const int64_t total = (int64_t)N * H * W;
  const int r = k / 2;
  for (int64_t q = (int64_t)blockIdx.x * NTHR + threadIdx.x; q < total; q += (int64_t)gridDim.x * NTHR) {
    const int w = (int)(q % W), h = (int)((q / W) % H);
    const int64_t base = q - (int64_t)h * W - w;  // first pixel of the frame
    float z = 0.f;
    for (int a = 0; a < k; ++a) {
      const int hh = h + a - r;
      if (hh < 0 || hh >= H) continue;
      for (int b = 0; b < k; ++b) {
        const int ww = w + b - r;
        if (ww < 0 || ww >= W) continue;
        const float* p = pooled + (base + (int64_t)hh * W + ww) * 2;
        z = fmaf(s_w[a * k + b], p[0], z);
        z = fmaf(s_w[k * k + a * k + b], p[1], z);
      }
    }
    gs[q] = 1.f / (1.f + __expf(-z));
  }
}

// out = relu(x * gc * gs)
template <typename T>
__global__ __launch_bounds__(NTHR) void cbam_apply_kernel(const T* __restrict__ x, const float* __restrict__ gc,
                                                          const float* __restrict__ gs, T* __restrict__ out, int64_t n8,
                                                          int P, int C8) {
  for (int64_t i = (int64_t)blockIdx.x * NTHR + threadIdx.x; i < n8; i += (int64_t)gridDim.x * NTHR) {
    const int64_t q = i / C8;
    const int cl = (int)(i - q * C8), n = (int)(q / P);
    const F8 v = load8(x + i * 8), g = load8(gc + ((size_t)n * C8 + cl) * 8);
    const float s = gs[q];
    F8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = fmaxf(v.v[k] * g.v[k] * s, 0.f);
    store8(out + i * 8, o);
  }
}

// ---- backward ----
// dpre[n][p] = (sum_c g * x * gc) * gs * (1 - gs), g = dout where x > 0: gradient at the spatial gate's pre-activation
template <typename T>
__global__ __launch_bounds__(NTHR) void bwd_spatial_sum_kernel(const T* __restrict__ x, const T* __restrict__ dout,
                                                               const float* __restrict__ gc, const float* __restrict__ gs,
                                                               float* __restrict__ dpre, int64_t NP, int P, int C) {
  const int C8 = C >> 3, PPB = NTHR / C8;
  const int cl = threadIdx.x % C8, pl = threadIdx.x / C8;
  for (int64_t q0 = (int64_t)blockIdx.x * PPB; q0 < NP; q0 += (int64_t)gridDim.x * PPB) {
    const int64_t q = q0 + pl;
    float s = 0.f;
    if (q < NP) {
      const int n = (int)(q / P);
      const F8 v = load8(x + q * C + cl * 8), d = load8(dout + q * C + cl * 8), g = load8(gc + (size_t)n * C + cl * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v.v[i] > 0.f ? d.v[i] * v.v[i] * g.v[i] : 0.f;
    }
    for (int o = 1; o < C8; o <<= 1) s += __shfl_xor(s, o, 64);
    if (q < NP && cl == 0) {
      const float t = gs[q];
      dpre[q] = s * t * (1.f - t);
    }
  }
}

// Transposed k x k convolution of dpre (gradient of the two pooled planes) and the weight gradient of the spatial
// gate's convolution.  A block walks tiles of TH x TW pixels; dpre and the pooled planes of a tile sit in LDS with a
// halo; thread (weight index) < 2*k*k keeps that weight's sum in a register over all of the block's tiles.
constexpr int TH = 8, TW = 32;
__global__ __launch_bounds__(NTHR) void bwd_spatial_conv_kernel(const float* __restrict__ dpre,
                                                                const float* __restrict__ pooled,
                                                                const float* __restrict__ Ws, float* __restrict__ dpooled,
                                                                float* __restrict__ dWs, int N, int H, int W, int k) {
  constexpr int HH = TH + 6, HW = TW + 6;
  __shared__ float s_d[HH][HW], s_p[2][HH][HW], s_w[2 * 49];
  const int r = k / 2;
  for (int i = threadIdx.x; i < 2 * k * k; i += NTHR) s_w[i] = Ws[i];
  const int tiles_h = (H + TH - 1) / TH, tiles_w = (W + TW - 1) / TW;
  const int64_t tiles = (int64_t)N * tiles_h * tiles_w;
  const int ty = threadIdx.x / TW, tx = threadIdx.x % TW;
  float wacc = 0.f;
  int wch = 0, wa = 0, wb = 0;
  const bool wthread = (int)threadIdx.x < 2 * k * k;
  if (wthread) {
    wch = threadIdx.x / (k * k);
    wa = (threadIdx.x % (k * k)) / k;
    wb = threadIdx.x % k;
  }
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int tw = (int)(t % tiles_w), th = (int)((t / tiles_w) % tiles_h), n = (int)(t / ((int64_t)tiles_w * tiles_h));
    const int h0 = th * TH, w0 = tw * TW;
    __syncthreads();
    for (int i = threadIdx.x; i < HH * HW; i += NTHR) {
      const int hh = h0 + i / HW - 3, ww = w0 + i % HW - 3;
      const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
      const int64_t q = ((int64_t)n * H + hh) * W + ww;
      s_d[i / HW][i % HW] = in ? dpre[q] : 0.f;
      s_p[0][i / HW][i % HW] = in ? pooled[q * 2] : 0.f;
      s_p[1][i / HW][i % HW] = in ? pooled[q * 2 + 1] : 0.f;
    }
    __syncthreads();
    const int h = h0 + ty, w = w0 + tx;
    if (h < H && w < W) {
      // y[q] = sum_{a,b} w[a][b] in[q + (a - r, b - r)]  =>  din[q] = sum_{a,b} w[a][b] dy[q - (a - r, b - r)]
      float d0 = 0.f, d1 = 0.f;
      for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) {
          const float dv = s_d[ty + 3 - (a - r)][tx + 3 - (b - r)];
          d0 = fmaf(s_w[a * k + b], dv, d0);
          d1 = fmaf(s_w[k * k + a * k + b], dv, d1);
        }
      const int64_t q = ((int64_t)n * H + h) * W + w;
      dpooled[q * 2] = d0;
      dpooled[q * 2 + 1] = d1;
    }
    if (wthread) {  // dW[ch][a][b] += sum over the tile's pixels dpre[q] * in[ch][q + (a - r, b - r)]
      for (int y = 0; y < TH; ++y)
        for (int x2 = 0; x2 < TW; ++x2) wacc = fmaf(s_d[y + 3][x2 + 3], s_p[wch][y + 3 + wa - r][x2 + 3 + wb - r], wacc);
    }
  }
  if (wthread) atomicAdd(dWs + threadIdx.x, wacc);
}

// dmid[c] = g*gs + dmean/C + [c == argc] dmax: gradient of x*gc.  Writes dx = dmid * gc and accumulates
// dgc[n][c] += sum_p dmid * x.  grid (chunks, N)
template <typename T>
__global__ __launch_bounds__(NTHR) void bwd_main_kernel(const T* __restrict__ x, const T* __restrict__ dout,
                                                        const float* __restrict__ gc, const float* __restrict__ gs,
                                                        const float* __restrict__ dpooled, const int* __restrict__ argc,
                                                        T* __restrict__ dx, float* __restrict__ dgc, int P, int C,
                                                        int rows_per_chunk) {
  __shared__ float red[NTHR][8];
  const int C8 = C >> 3, RL = NTHR / C8;
  const int cl = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int n = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(P, r0 + rows_per_chunk);
  const float invC = 1.f / (float)C;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (rl < RL) {
    const F8 g = load8(gc + (size_t)n * C + cl * 8);
    for (int r = r0 + rl; r < r1; r += RL) {
      const int64_t q = (int64_t)n * P + r;
      const F8 v = load8(x + q * C + cl * 8), d = load8(dout + q * C + cl * 8);
      const float s = gs[q], dm = dpooled[q * 2] * invC, dM = dpooled[q * 2 + 1];
      const int am = argc[q] - cl * 8;
      F8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float t = (v.v[i] > 0.f ? d.v[i] * s : 0.f) + dm;
        if (i == am) t += dM;
        acc[i] = fmaf(t, v.v[i], acc[i]);
        o.v[i] = t * g.v[i];
      }
      store8(dx + q * C + cl * 8, o);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NTHR) {
    float a = 0.f;
    for (int k = 0; k < RL; ++k) a += red[k * C8 + (c >> 3)][c & 7];
    atomicAdd(dgc + (size_t)n * C + c, a);
  }
}

// grid N: backward of the channel gate's two-layer map; davg / dmx per frame, dW1 / dW2 with atomics
__global__ __launch_bounds__(NTHR) void chan_gate_bwd_kernel(const float* __restrict__ dgc, const float* __restrict__ gate,
                                                             const float* __restrict__ avg, const float* __restrict__ mx,
                                                             const float* __restrict__ hid, const float* __restrict__ W1,
                                                             const float* __restrict__ W2, float* __restrict__ davg,
                                                             float* __restrict__ dmx, float* __restrict__ dW1,
                                                             float* __restrict__ dW2, int C, int Ch) {
  __shared__ float s_dz[512], s_dh[2][32], s_h[2][32];
  const int n = blockIdx.x;
  if ((int)threadIdx.x < 2 * Ch) s_h[threadIdx.x / Ch][threadIdx.x % Ch] = hid[(size_t)n * 2 * Ch + threadIdx.x];
  for (int c = threadIdx.x; c < C; c += NTHR) {
    const float g = gate[(size_t)n * C + c];
    s_dz[c] = dgc[(size_t)n * C + c] * g * (1.f - g);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NTHR)
    for (int j = 0; j < Ch; ++j) atomicAdd(dW2 + c * Ch + j, s_dz[c] * (s_h[0][j] + s_h[1][j]));
  if ((int)threadIdx.x < 2 * Ch) {
    const int which = threadIdx.x / Ch, j = threadIdx.x % Ch;
    float d = 0.f;
    for (int c = 0; c < C; ++c) d = fmaf(W2[c * Ch + j], s_dz[c], d);
    s_dh[which][j] = s_h[which][j] > 0.f ? d : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NTHR) {
    const float a = avg[(size_t)n * C + c], m = mx[(size_t)n * C + c];
    float da = 0.f, dm = 0.f;
    for (int j = 0; j < Ch; ++j) {
      da = fmaf(W1[j * C + c], s_dh[0][j], da);
      dm = fmaf(W1[j * C + c], s_dh[1][j], dm);
      atomicAdd(dW1 + j * C + c, s_dh[0][j] * a + s_dh[1][j] * m);
    }
    davg[(size_t)n * C + c] = da;
    dmx[(size_t)n * C + c] = dm;
  }
}

// dx += davg[n][c] / P + [p == argp[n][c]] dmx[n][c]: the pooled inputs of the channel gate
template <typename T>
__global__ __launch_bounds__(NTHR) void bwd_pool_kernel(T* __restrict__ dx, const float* __restrict__ davg,
                                                        const float* __restrict__ dmx, const int* __restrict__ argp,
                                                        int64_t n8, int P, int C8) {
  const float invP = 1.f / (float)P;
  for (int64_t i = (int64_t)blockIdx.x * NTHR + threadIdx.x; i < n8; i += (int64_t)gridDim.x * NTHR) {
    const int64_t q = i / C8;
    const int cl = (int)(i - q * C8), n = (int)(q / P), p = (int)(q - (int64_t)n * P);
    const size_t o = ((size_t)n * C8 + cl) * 8;
    F8 v = load8(dx + i * 8);
    const F8 a = load8(davg + o), m = load8(dmx + o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v.v[k] = fmaf(a.v[k], invP, v.v[k]);
      if (argp[o + k] == p) v.v[k] += m.v[k];
    }
    store8(dx + i * 8, v);
  }
}

int ew_blocks(int64_t work) {
  const int64_t b = ceil_div64(work, NTHR);
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int chunks_for(int P) {  // row chunks per frame of the (chunks, N) kernels
  int c = ceil_div(P, 2048);
  return c < 1 ? 1 : (c > 64 ? 64 : c);
}

bool cbam_shape_ok(int N, int H, int W, int C, int Ch, int k) {
  const int C8 = C / 8;
  return N > 0 && H > 0 && W > 0 && C % 8 == 0 && C8 >= 1 && C8 <= 64 && (C8 & (C8 - 1)) == 0 && Ch >= 1 && Ch <= 32 &&
         (k == 3 || k == 5 || k == 7) && (int64_t)N * H * W * C < (1ll << 31);
}

}  // namespace

#define DISPATCH_T(dtype, who, ...)                          \
  if (dtype == VC_BF16) { typedef bf16 T; __VA_ARGS__ }      \
  else if (dtype == VC_F32) { typedef float T; __VA_ARGS__ } \
  else { vc_set_error("%s: unsupported dtype %d", who, dtype); return VC_ERR_DTYPE; }

extern "C" {

// floats of workspace vc_cbam_fwd / vc_cbam_bwd need (the partial pooling results of the frames' row chunks)
int64_t vc_cbam_workspace_floats(int N, int H, int W, int C) { return (int64_t)3 * N * chunks_for(H * W) * C; }

// x [N][H*W][C] -> out = relu(x * gc * gs).  Saved for backward (all fp32 / int32, caller-allocated):
// avg, mx [N][C]; argp [N][C] (pixel of the maximum); hid [N][2][Ch]; gc [N][C]; pooled [N][H*W][2];
// argc [N][H*W] (channel of the maximum); gs [N][H*W].
int vc_cbam_fwd(int dtype, const void* x, const float* W1, const float* W2, const float* Ws, void* out, float* avg,
                float* mx, int* argp, float* hid, float* gc, float* pooled, int* argc, float* gs, float* workspace,
                int N, int H, int W, int C, int Ch, int k, hipStream_t stream) {
  VC_CHECK_ARG(x && W1 && W2 && Ws && out && avg && mx && argp && hid && gc && pooled && argc && gs && workspace,
               "vc_cbam_fwd: null pointer");
  VC_CHECK_SHAPE(cbam_shape_ok(N, H, W, C, Ch, k), "vc_cbam_fwd: unsupported shape (C/8 a power of two <= 64, C/r <= 32, "
                 "k in 3/5/7): N=%d H=%d W=%d C=%d Ch=%d k=%d", N, H, W, C, Ch, k);
  const int P = H * W, chunks = chunks_for(P), rpc = ceil_div(P, chunks);
  float* psum = workspace;
  float* pmax = psum + (size_t)N * chunks * C;
  int* parg = reinterpret_cast<int*>(pmax + (size_t)N * chunks * C);
  const int64_t NP = (int64_t)N * P, n8 = NP * (C / 8);
  DISPATCH_T(dtype, "vc_cbam_fwd",
    hipLaunchKernelGGL((chan_pool_partial_kernel<T>), dim3(chunks, N), dim3(NTHR), 0, stream, (const T*)x, psum, pmax,
                       parg, P, C, rpc);
    hipLaunchKernelGGL(chan_gate_kernel, dim3(N), dim3(NTHR), 0, stream, psum, pmax, parg, W1, W2, avg, mx, argp, hid,
                       gc, chunks, P, C, Ch);
    hipLaunchKernelGGL((spatial_pool_kernel<T>), dim3(ew_blocks(n8)), dim3(NTHR), 0, stream, (const T*)x, gc, pooled,
                       argc, NP, P, C);
    hipLaunchKernelGGL(spatial_gate_kernel, dim3(ew_blocks(NP)), dim3(NTHR), 0, stream, pooled, Ws, gs, N, H, W, k);
    hipLaunchKernelGGL((cbam_apply_kernel<T>), dim3(ew_blocks(n8)), dim3(NTHR), 0, stream, (const T*)x, gc, gs, (T*)out,
                       n8, P, C / 8);)
  VC_LAUNCH_CHECK("vc_cbam_fwd");
  return VC_OK;
}

// dx [N][H*W][C] (every element written); dW1 [Ch][C], dW2 [C][Ch], dWs [2][k][k] are ACCUMULATED into (fp32).
// scratch: dpre [N][H*W], dpooled [N][H*W][2], dgc [N][C] (zeroed here), davg, dmx [N][C] -- caller-allocated fp32.
int vc_cbam_bwd(int dtype, const void* x, const void* dout, const float* W1, const float* W2, const float* Ws,
                const float* avg, const float* mx, const int* argp, const float* hid, const float* gc,
                const float* pooled, const int* argc, const float* gs, void* dx, float* dW1, float* dW2, float* dWs,
                float* dpre, float* dpooled, float* dgc, float* davg, float* dmx, int N, int H, int W, int C, int Ch,
                int k, hipStream_t stream) {
  VC_CHECK_ARG(x && dout && W1 && W2 && Ws && avg && mx && argp && hid && gc && pooled && argc && gs && dx && dW1 &&
               dW2 && dWs && dpre && dpooled && dgc && davg && dmx, "vc_cbam_bwd: null pointer");
  VC_CHECK_SHAPE(cbam_shape_ok(N, H, W, C, Ch, k), "vc_cbam_bwd: unsupported shape");
  const int P = H * W, chunks = chunks_for(P), rpc = ceil_div(P, chunks);
  const int64_t NP = (int64_t)N * P, n8 = NP * (C / 8);
  const int64_t tiles = (int64_t)N * ceil_div(H, TH) * ceil_div(W, TW);
  if (hipMemsetAsync(dgc, 0, (size_t)N * C * sizeof(float), stream) != hipSuccess) {
    vc_set_error("vc_cbam_bwd: clearing the channel-gate sums failed");
    return VC_ERR_HIP;
  }
  DISPATCH_T(dtype, "vc_cbam_bwd",
    hipLaunchKernelGGL((bwd_spatial_sum_kernel<T>), dim3(ew_blocks(n8)), dim3(NTHR), 0, stream, (const T*)x,
                       (const T*)dout, gc, gs, dpre, NP, P, C);
    hipLaunchKernelGGL(bwd_spatial_conv_kernel, dim3((unsigned)(tiles < 1024 ? tiles : 1024)), dim3(NTHR), 0, stream,
                       dpre, pooled, Ws, dpooled, dWs, N, H, W, k);
    hipLaunchKernelGGL((bwd_main_kernel<T>), dim3(chunks, N), dim3(NTHR), 0, stream, (const T*)x, (const T*)dout, gc, gs,
                       dpooled, argc, (T*)dx, dgc, P, C, rpc);
    hipLaunchKernelGGL(chan_gate_bwd_kernel, dim3(N), dim3(NTHR), 0, stream, dgc, gc, avg, mx, hid, W1, W2, davg, dmx,
                       dW1, dW2, C, Ch);
    hipLaunchKernelGGL((bwd_pool_kernel<T>), dim3(ew_blocks(n8)), dim3(NTHR), 0, stream, (T*)dx, davg, dmx, argp, n8, P,
                       C / 8);)
  VC_LAUNCH_CHECK("vc_cbam_bwd");
  return VC_OK;
}

}  // extern "C"
