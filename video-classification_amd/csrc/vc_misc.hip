// Bandwidth-bound kernels around the contractions (gfx950): clip layout conversion, pooling,
// small elementwise ops, the LSTM cell, the fused cross-entropy head and the fused optimizer.
// Channels-last everywhere; 8 channels (16 B of bf16 / 32 B of fp32) per lane per access.
#include "vc_common.h"

namespace {

constexpr int NTHR = 256;

inline int ew_grid(int64_t n) {
  int64_t b = ceil_div64(n, NTHR);
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// ---------------------------------------------------------------------------
// Clip layout: fp32 [B][T][C][S] (frames NCHW, S = H*W)  ->  T [F][S][Cp], zero-padded
// channels.  time_major: output frame f = t*B + b (BatchNorm groups contiguous);
// otherwise f = b*T + t (Conv3D: [B][C][T][S] -> [B][T][S][Cp] uses chan_outer = 1).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void clip_to_cl_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int Tn, int C, int S,
                                  int Cp, int time_major, int chan_outer) {
  const int64_t total = (int64_t)B * Tn * S;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(idx % S);
    const int f = (int)(idx / S);
    int b, t;
    if (time_major) {
      t = f / B;
      b = f - t * B;
    } else {
      b = f / Tn;
      t = f - b * Tn;
    }
    T* o = out + idx * Cp;
    for (int c = 0; c < Cp; ++c) {
      float v = 0.f;
      if (c < C) {
        const size_t src = chan_outer ? (((size_t)b * C + c) * Tn + t) * S + s : (((size_t)b * Tn + t) * C + c) * S + s;
        v = x[src];
      }
      o[c] = from_f32<T>(v);
    }
  }
}

// ---------------------------------------------------------------------------
// Width-unfolded input of a network's first convolution (1 or 3 input channels): instead of padding the
// channels to 8 (5/8 or 7/8 of every product wasted), the kw taps along W are folded into the channel axis,
//   out[f][h][wo][j*C + c] = x[.., c, h, wo*sw - pw + j]      (zero outside the image and for j*C + c >= kw*C),
// so the layer becomes a (kd, kh, 1) convolution with stride (sd, sh, 1) over Jp = pad8(kw*C) channels.
// Source layouts and frame order as for clip_to_cl_kernel.
// ---------------------------------------------------------------------------
// One block per `rpb` consecutive (frame, image row) pairs: their C source rows each are read coalesced into LDS
// first (all loads of the block in flight together: with one row per block a CU held 21 KiB of requests and the pass
// ran at half the copy bandwidth), then every thread assembles the Jp values of its output pixels from LDS and writes
// them as 16-byte vectors.
template <typename T>
__global__ void clip_unfold_w_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int Tn, int C, int H,
                                     int W, int Wo, int kw, int sw, int pw, int Jp, int time_major, int chan_outer,
                                     int rows, int rpb) {
  extern __shared__ __attribute__((aligned(16))) float srow[];  // [rpb][C][W]
  const int S = H * W;
  const int fh0 = blockIdx.x * rpb;
  const int nr = min(rpb, rows - fh0);
  for (int rc = 0; rc < nr * C; ++rc) {  // one source row per (image row, channel), 16-byte loads where it allows
    const int r = rc / C, c = rc - r * C;
    const int fh = fh0 + r;
    const int h = fh % H, f = fh / H;
    int b, t;
    if (time_major) {
      t = f / B;
      b = f - t * B;
    } else {
      b = f / Tn;
      t = f - b * Tn;
    }
    const float* row = x + (chan_outer ? (((size_t)b * C + c) * Tn + t) * S : (((size_t)b * Tn + t) * C + c) * S) +
                       (size_t)h * W;
    float* dst = srow + rc * W;
    if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
      for (int i = threadIdx.x; i < (W >> 2); i += blockDim.x)
        reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(row)[i];
    } else {
      for (int i = threadIdx.x; i < W; i += blockDim.x) dst[i] = row[i];
    }
  }
  __syncthreads();
  const int J8 = Jp >> 3, kwc = kw * C;
  for (int r = 0; r < nr; ++r) {
    const int fh = fh0 + r;
    const float* sr = srow + r * C * W;
    // one 8-channel chunk of one output pixel per step; the (tap, channel) of the chunk's first element is found by
    // counting, so that the loop has no integer divisions
    for (int wo = threadIdx.x / J8, ch = threadIdx.x - wo * J8; wo < Wo;) {
      int q = ch * 8, j = q / C, c = q - j * C;
      const int w0 = wo * sw - pw;
      F8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e, ++q) {
        float val = 0.f;
        if (q < kwc && (unsigned)(w0 + j) < (unsigned)W) val = sr[c * W + w0 + j];
        v.v[e] = val;
        if (++c == C) {
          c = 0;
          ++j;
        }
      }
      store8(out + ((size_t)fh * Wo + wo) * Jp + ch * 8, v);
      // advance by blockDim.x chunks
      ch += blockDim.x % J8;
      wo += blockDim.x / J8;
      if (ch >= J8) {
        ch -= J8;
        ++wo;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Max pooling over (H, W) of [N][H][W][C]; records the winning tap per output element.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int N,
                                   int H, int W, int C8, int Ho, int Wo, int k, int s, int p) {
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % C8);
    int64_t t = i / C8;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float best[8];
    int bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      best[j] = -INFINITY;
      bi[j] = 0;
    }
    for (int r = 0; r < k; ++r) {
      const int ih = oh * s - p + r;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int q = 0; q < k; ++q) {
        const int iw = ow * s - p + q;
        if ((unsigned)iw >= (unsigned)W) continue;
        const F8 v = load8(x + ((((size_t)n * H + ih) * W + iw) * C8 + cc) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (v.v[j] > best[j]) {
            best[j] = v.v[j];
            bi[j] = r * k + q;
          }
      }
    }
    F8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = best[j];
    store8(y + i * 8, o);
    if (idx) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo |= (uint32_t)bi[j] << (8 * j);
        hi |= (uint32_t)bi[4 + j] << (8 * j);
      }
      *reinterpret_cast<uint2*>(idx + i * 8) = make_uint2(lo, hi);
    }
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int N,
                                   int H, int W, int C8, int Ho, int Wo, int k, int s, int p) {
  // one block row per (sample, image row): the only division left is the one that splits a row item into
  // (column, channel chunk)
  const int ih = blockIdx.y, n = blockIdx.z;
  const int oh_lo = max(0, (ih + p - k + s) / s), oh_hi = min(Ho - 1, (ih + p) / s);
  for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < W * C8; it += gridDim.x * blockDim.x) {
    const int iw = it / C8, cc = it - iw * C8;
    const int64_t i = ((int64_t)n * H + ih) * W * C8 + it;
    F8 acc;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc.v[j] = 0.f;
    // output windows that contain (ih, iw): oh*s - p <= ih <= oh*s - p + k - 1
    const int ow_lo = max(0, (iw + p - k + s) / s), ow_hi = min(Wo - 1, (iw + p) / s);
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int r = ih - (oh * s - p);
      if (r < 0 || r >= k) continue;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int q = iw - (ow * s - p);
        if (q < 0 || q >= k) continue;
        const size_t o = ((((size_t)n * Ho + oh) * Wo + ow) * C8 + cc) * 8;
        const uint2 w = *reinterpret_cast<const uint2*>(idx + o);
        const F8 d = load8(dy + o);
        const int tap = r * k + q;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int b = (int)(((j < 4 ? w.x : w.y) >> (8 * (j & 3))) & 0xff);
          if (b == tap) acc.v[j] += d.v[j];
        }
      }
    }
    store8(dx + i * 8, acc);
  }
}

// The ResNet stem's pool (3 x 3, stride 2, padding 1, even H and W) backward: a thread owns the 2 x 2 input pixels
// (2a .. 2a+1, 2b .. 2b+1) of one 8-channel chunk.  They lie in the four windows (a, b), (a, b+1), (a+1, b), (a+1, b+1)
// only, each read once (the per-pixel form above reads 9 windows for the same four pixels), and every pixel's taps in
// those windows are constants: (2a, 2b) is tap 4 of (a, b); (2a, 2b+1) taps 5 / 3 of (a, b) / (a, b+1); (2a+1, 2b) taps
// 7 / 1 of (a, b) / (a+1, b); (2a+1, 2b+1) taps 8 / 6 / 2 / 0 of the four.
template <typename T>
__global__ void maxpool_bwd_3s2p1_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx,
                                         int N, int Ho, int Wo, int C8) {
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  const int H = 2 * Ho, W = 2 * Wo;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(it % C8);
    int64_t t = it / C8;
    const int b = (int)(t % Wo);
    t /= Wo;
    const int a = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const bool hb = b + 1 < Wo, ha = a + 1 < Ho;
    const size_t o00 = ((((size_t)n * Ho + a) * Wo + b) * C8 + cc) * 8;
    const size_t o01 = o00 + (size_t)C8 * 8, o10 = o00 + (size_t)Wo * C8 * 8, o11 = o10 + (size_t)C8 * 8;
    F8 d[4];
    uint2 w[4];
    const F8 zero = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    const uint2 none = {0xFFFFFFFFu, 0xFFFFFFFFu};  // no tap has index 255
    d[0] = load8(dy + o00);
    w[0] = *reinterpret_cast<const uint2*>(idx + o00);
    d[1] = hb ? load8(dy + o01) : zero;
    w[1] = hb ? *reinterpret_cast<const uint2*>(idx + o01) : none;
    d[2] = ha ? load8(dy + o10) : zero;
    w[2] = ha ? *reinterpret_cast<const uint2*>(idx + o10) : none;
    d[3] = (ha && hb) ? load8(dy + o11) : zero;
    w[3] = (ha && hb) ? *reinterpret_cast<const uint2*>(idx + o11) : none;
    F8 p00, p01, p10, p11;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int tp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) tp[k] = (int)(((j < 4 ? w[k].x : w[k].y) >> (8 * (j & 3))) & 0xff);
      p00.v[j] = tp[0] == 4 ? d[0].v[j] : 0.f;
      p01.v[j] = (tp[0] == 5 ? d[0].v[j] : 0.f) + (tp[1] == 3 ? d[1].v[j] : 0.f);
      p10.v[j] = (tp[0] == 7 ? d[0].v[j] : 0.f) + (tp[2] == 1 ? d[2].v[j] : 0.f);
      // the same order of additions as the per-pixel form (windows by row, then by column)
      p11.v[j] = (((tp[0] == 8 ? d[0].v[j] : 0.f) + (tp[1] == 6 ? d[1].v[j] : 0.f)) + (tp[2] == 2 ? d[2].v[j] : 0.f)) +
                 (tp[3] == 0 ? d[3].v[j] : 0.f);
    }
    const size_t i00 = ((((size_t)n * H + 2 * a) * W + 2 * b) * C8 + cc) * 8;
    store8(dx + i00, p00);
    store8(dx + i00 + (size_t)C8 * 8, p01);
    store8(dx + i00 + (size_t)W * C8 * 8, p10);
    store8(dx + i00 + (size_t)W * C8 * 8 + (size_t)C8 * 8, p11);
  }
}

// global average pool over S positions: [N][S][C] -> [N][C]
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int S, int C8) {
  const int64_t total = (int64_t)N * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % C8);
    const int n = (int)(i / C8);
    F8 acc;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc.v[j] = 0.f;
    for (int s = 0; s < S; ++s) {
      const F8 v = load8(x + (((size_t)n * S + s) * C8 + cc) * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc.v[j] += v.v[j];
    }
    const float inv = 1.f / S;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc.v[j] *= inv;
    store8(y + i * 8, acc);
  }
}

// long pooling extents (Conv3D: 16 x 7 x 7 = 784 positions, few samples): one wave per (sample, 8 channels),
// the positions spread over its lanes, instead of one thread walking all of them
template <typename T>
__global__ void avgpool_fwd_wave_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int S, int C8) {
  const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pair >= (int64_t)N * C8) return;
  const int lane = threadIdx.x & 63;
  const int cc = (int)(pair % C8);
  const int n = (int)(pair / C8);
  F8 acc;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc.v[j] = 0.f;
  for (int sp = lane; sp < S; sp += 64) {
    const F8 v = load8(x + (((size_t)n * S + sp) * C8 + cc) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc.v[j] += v.v[j];
  }
  const float inv = 1.f / S;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc.v[j] = wave_sum(acc.v[j]) * inv;
  if (lane == 0) store8(y + pair * 8, acc);
}

template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int S, int C8) {
  const int64_t total = (int64_t)N * S * C8;
  const float inv = 1.f / S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % C8);
    const int n = (int)(i / ((int64_t)S * C8));
    F8 d = load8(dy + ((size_t)n * C8 + cc) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) d.v[j] *= inv;
    store8(dx + i * 8, d);
  }
}

// ---------------------------------------------------------------------------
// Small elementwise ops on [n8 * 8] elements.
// OP 0: out = a + b              OP 1: out = a * (b > 0)   (ReLU backward: a = dout, b = act)
// OP 2: out = relu(a)            OP 3: out = gelu(a)        OP 4: out = a * gelu'(b)
// OP 5: out = sigmoid(a)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return vc_gelu(x); }
__device__ __forceinline__ float gelu_grad_f(float x) { return vc_gelu_grad(x); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T, int OP>
__global__ void ew_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    F8 av = load8(a + i * 8), bv;
    if (OP == 0 || OP == 1 || OP == 4) bv = load8(b + i * 8);
    F8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (OP == 0) o.v[j] = av.v[j] + bv.v[j];
      if (OP == 1) o.v[j] = bv.v[j] > 0.f ? av.v[j] : 0.f;
      if (OP == 2) o.v[j] = fmaxf(av.v[j], 0.f);
      if (OP == 3) o.v[j] = gelu_f(av.v[j]);
      if (OP == 4) o.v[j] = av.v[j] * gelu_grad_f(bv.v[j]);
      if (OP == 5) o.v[j] = sigmoid_f(av.v[j]);
    }
    store8(out + i * 8, o);
  }
}

// dtype conversion: fp32 <-> T
template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ a, TO* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = from_f32<TO>(to_f32(a[i]));
}

// column sums of [M][C] (bias gradients): out[c] += sum_m a[m][c]
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ a, float* __restrict__ out, int M, int C, int lda,
                              int rows_per_block, size_t slot_stride) {
  // slot_stride != 0 (deterministic mode): the block of row range y adds into its own copy y of out
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) acc += to_f32(a[(size_t)r * lda + c]);
  atomicAdd(out + (size_t)blockIdx.y * slot_stride + c, acc);
}

// vectorised variant for dense rows with C % 8 == 0: 8 channels per lane, the block's row lanes are
// reduced through LDS, one atomic per channel per block
template <typename T>
__global__ void colsum_vec_kernel(const T* __restrict__ a, float* __restrict__ out, int M, int C, int Cvalid,
                                  int rows_per_block, size_t slot_stride) {
  __shared__ float red[NTHR][8];
  const int C8 = C >> 3;
  const int lanesC = C8 < NTHR ? C8 : NTHR;
  const int RL = NTHR / lanesC;
  const int tid = threadIdx.x;
  const int cl = tid % lanesC, rl = tid / lanesC;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(M, r_begin + rows_per_block);
  for (int cc0 = 0; cc0 < C8; cc0 += lanesC) {
    const int cc = cc0 + cl;
    const bool cvalid = cc < C8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    if (rl < RL && cvalid) {
#pragma unroll 4
      for (int r = r_begin + rl; r < r_end; r += RL) {
        const F8 v = load8(a + (size_t)r * C + cc * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v.v[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[tid][i] = acc[i];
    __syncthreads();
    if (rl == 0 && cvalid) {
      for (int k = 1; k < RL; ++k) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += red[k * lanesC + cl][i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (cc * 8 + i < Cvalid) atomicAdd(out + (size_t)blockIdx.x * slot_stride + cc * 8 + i, acc[i]);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// LSTM cell (PyTorch gate order i, f, g, o).  gates_x already holds x W_ih^T + b_ih + b_hh;
// gates_h (nullable at the first step: zero state) holds h_{t-1} W_hh^T.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ gates_x, int ldx, const float* __restrict__ gates_h,
                                     const float* __restrict__ c_prev, T* __restrict__ h_out, int ldh,
                                     float* __restrict__ c_out, float* __restrict__ act_out, int B, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i - b * H;
  float pre[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pre[k] = gates_x[(size_t)b * ldx + k * H + j];
    if (gates_h) pre[k] += gates_h[(size_t)b * 4 * H + k * H + j];
  }
  const float ig = sigmoid_f(pre[0]), fg = sigmoid_f(pre[1]), gg = tanhf(pre[2]), og = sigmoid_f(pre[3]);
  const float cp = c_prev ? c_prev[i] : 0.f;
  const float c = fg * cp + ig * gg;
  const float h = og * tanhf(c);
  c_out[i] = c;
  h_out[(size_t)b * ldh + j] = from_f32<T>(h);
  float* a = act_out + (size_t)b * 4 * H + j;
  a[0] = ig;
  a[H] = fg;
  a[2 * H] = gg;
  a[3 * H] = og;
}

// dh = dh_out (gradient from the layer above / output at this step, nullable) + dh_rec (from step t+1, nullable)
template <typename T>
__global__ void lstm_cell_bwd_kernel(const T* __restrict__ dh_out, int ldo, const float* __restrict__ dh_rec,
                                     const float* __restrict__ dc_next, const float* __restrict__ act,
                                     const float* __restrict__ c, const float* __restrict__ c_prev,
                                     T* __restrict__ dgates, float* __restrict__ dc_prev, int B, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i - b * H;
  float dh = 0.f;
  if (dh_out) dh += to_f32(dh_out[(size_t)b * ldo + j]);
  if (dh_rec) dh += dh_rec[i];
  const float* a = act + (size_t)b * 4 * H + j;
  const float ig = a[0], fg = a[H], gg = a[2 * H], og = a[3 * H];
  const float tc = tanhf(c[i]);
  float dc = dh * og * (1.f - tc * tc);
  if (dc_next) dc += dc_next[i];
  const float cp = c_prev ? c_prev[i] : 0.f;
  const float d_o = dh * tc, d_i = dc * gg, d_g = dc * ig, d_f = dc * cp;
  dc_prev[i] = dc * fg;
  T* dg = dgates + (size_t)b * 4 * H + j;
  dg[0] = from_f32<T>(d_i * ig * (1.f - ig));
  dg[H] = from_f32<T>(d_f * fg * (1.f - fg));
  dg[2 * H] = from_f32<T>(d_g * (1.f - gg * gg));
  dg[3 * H] = from_f32<T>(d_o * og * (1.f - og));
}

// ---------------------------------------------------------------------------
// Cross-entropy head: one wave per row of fp32 logits [B][ld]; mean loss, argmax,
// number of correct predictions and dlogits = (softmax - onehot) * grad_scale.
// ---------------------------------------------------------------------------
template <typename TG>
__global__ void ce_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                          float* __restrict__ loss, int64_t* __restrict__ pred, int* __restrict__ correct,
                          TG* __restrict__ dlogits, int ldd, int B, int C, float loss_scale, float grad_scale,
                          int loss_per_row) {
  // loss_per_row (deterministic mode): loss[row] receives the row's term; the caller adds the rows in order
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* lr = logits + (size_t)row * ld;
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = lr[c];
    if (v > mx) {
      mx = v;
      mi = c;
    }
  }
  // wave arg-max with lowest index among equal maxima (torch.argmax on CPU returns the first)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (ov > mx || (ov == mx && oi < mi)) {
      mx = ov;
      mi = oi;
    }
  }
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += __expf(lr[c] - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se);
  const int tgt = (int)target[row];
  if (lane == 0) {
    atomicAdd(loss + (loss_per_row ? row : 0), (lse - lr[tgt]) * loss_scale);
    if (pred) pred[row] = mi;
    if (correct && mi == tgt) atomicAdd(correct, 1);
  }
  if (dlogits) {
    for (int c = lane; c < ldd; c += 64) {
      float g = 0.f;
      if (c < C) g = (__expf(lr[c] - lse) - (c == tgt ? 1.f : 0.f)) * grad_scale;
      dlogits[(size_t)row * ldd + c] = from_f32<TG>(g);
    }
  }
}

// ---------------------------------------------------------------------------
// Fused optimizer over one flat fp32 parameter buffer.
// mode 0: Adam with L2 weight decay folded into the gradient (torch.optim.Adam)
// mode 1: AdamW (decoupled decay)      mode 2: SGD with momentum (torch.optim.SGD, dampening 0)
// ---------------------------------------------------------------------------
__global__ void optim_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, int mode, float lr, float b1, float b2, float eps,
                             float wd, int step, float grad_scale, const float* __restrict__ lr_dev,
                             const int64_t* __restrict__ step_dev) {
  // hyper-parameters that change between steps may live in device memory so that a captured graph of the
  // training step replays with the current values
  if (lr_dev) lr = *lr_dev;
  if (step_dev) step = (int)*step_dev;
  const float bc1 = 1.f - (float)pow((double)b1, (double)step);
  const float bc2_sqrt = sqrtf(1.f - (float)pow((double)b2, (double)step));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pv = p[i];
    float gv = g[i] * grad_scale;
    if (mode == 2) {
      gv = fmaf(wd, pv, gv);
      // torch's first step sets buf = g; the buffer starts at zero, so b1 * 0 + g is the same value, and a
      // momentum buffer restored from a checkpoint is never overwritten
      const float buf = fmaf(b1, m[i], gv);
      m[i] = buf;
      p[i] = pv - lr * buf;
      continue;
    }
    if (mode == 0) gv = fmaf(wd, pv, gv);
    else pv *= (1.f - lr * wd);
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv - (lr / bc1) * (mv / denom);
  }
}

__global__ void counter_add_kernel(int64_t* c, int64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += inc;
}

}  // namespace

#define DISPATCH_T(dtype, who, ...)                              \
  if (dtype == VC_BF16) { typedef bf16 T; __VA_ARGS__ }          \
  else if (dtype == VC_F32) { typedef float T; __VA_ARGS__ }     \
  else { vc_set_error("%s: unsupported dtype %d", who, dtype); return VC_ERR_DTYPE; }

extern "C" {

int vc_clip_to_channels_last(int dtype, const float* x, void* out, int B, int T_, int C, int S, int Cp, int time_major,
                             int chan_outer, hipStream_t stream) {
  VC_CHECK_ARG(x && out, "vc_clip_to_channels_last: null pointer");
  VC_CHECK_SHAPE(B > 0 && T_ > 0 && C > 0 && S > 0 && Cp >= C, "vc_clip_to_channels_last: bad shape");
  const int64_t total = (int64_t)B * T_ * S;
  DISPATCH_T(dtype, "vc_clip_to_channels_last",
             hipLaunchKernelGGL((clip_to_cl_kernel<T>), dim3(ew_grid(total)), dim3(NTHR), 0, stream, x, (T*)out, B, T_,
                                C, S, Cp, time_major, chan_outer);)
  VC_LAUNCH_CHECK("vc_clip_to_channels_last");
  return VC_OK;
}

int vc_clip_unfold_w(int dtype, const float* x, void* out, int B, int T_, int C, int H, int W, int kw, int sw, int pw,
                     int Jp, int time_major, int chan_outer, hipStream_t stream) {
  VC_CHECK_ARG(x && out, "vc_clip_unfold_w: null pointer");
  VC_CHECK_SHAPE(B > 0 && T_ > 0 && C > 0 && H > 0 && W > 0 && kw > 0 && sw > 0 && pw >= 0 && Jp >= kw * C &&
                     Jp % 8 == 0 && W + 2 * pw >= kw,
                 "vc_clip_unfold_w: bad shape");
  const int Wo = (W + 2 * pw - kw) / sw + 1;
  const int64_t rows = (int64_t)B * T_ * H;
  VC_CHECK_SHAPE(rows < (1ll << 31) && (size_t)C * W * sizeof(float) <= 48 * 1024, "vc_clip_unfold_w: image too large");
  static const int rpb_env = [] {
    const char* e = getenv("VC_UNFOLD_ROWS");
    const int n = e ? atoi(e) : 0;
    return n > 0 ? n : 4;
  }();
  int rpb = rpb_env;  // image rows per block: as many as 16 KiB of LDS hold (eight blocks per CU), at most rpb_env
  while (rpb > 1 && (size_t)rpb * C * W * sizeof(float) > 16 * 1024) rpb >>= 1;
  const size_t lds = (size_t)rpb * C * W * sizeof(float);
  const unsigned blocks = (unsigned)((rows + rpb - 1) / rpb);
  DISPATCH_T(dtype, "vc_clip_unfold_w",
             hipLaunchKernelGGL((clip_unfold_w_kernel<T>), dim3(blocks), dim3(NTHR), lds, stream, x, (T*)out, B, T_, C, H,
                                W, Wo, kw, sw, pw, Jp, time_major, chan_outer, (int)rows, rpb);)
  VC_LAUNCH_CHECK("vc_clip_unfold_w");
  return VC_OK;
}

int vc_maxpool2d_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int k, int s, int p,
                     hipStream_t stream) {
  VC_CHECK_ARG(x && y, "vc_maxpool2d_fwd: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0 && k > 0 && k * k <= 255 && s > 0 && p >= 0 && p < k, "vc_maxpool2d_fwd: bad shape");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  const int64_t total = (int64_t)N * Ho * Wo * (C / 8);
  DISPATCH_T(dtype, "vc_maxpool2d_fwd",
             hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(ew_grid(total)), dim3(NTHR), 0, stream, (const T*)x,
                                (T*)y, idx, N, H, W, C / 8, Ho, Wo, k, s, p);)
  VC_LAUNCH_CHECK("vc_maxpool2d_fwd");
  return VC_OK;
}

int vc_maxpool2d_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int k, int s,
                     int p, hipStream_t stream) {
  VC_CHECK_ARG(dy && idx && dx, "vc_maxpool2d_bwd: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0 && k > 0 && s > 0 && p >= 0 && p < k, "vc_maxpool2d_bwd: bad shape");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  VC_CHECK_SHAPE(H <= 65535 && N <= 65535, "vc_maxpool2d_bwd: at most 65535 rows and samples");
  if (k == 3 && s == 2 && p == 1 && H % 2 == 0 && W % 2 == 0) {
    const int64_t total = (int64_t)N * Ho * Wo * (C / 8);
    DISPATCH_T(dtype, "vc_maxpool2d_bwd",
               hipLaunchKernelGGL((maxpool_bwd_3s2p1_kernel<T>), dim3(ew_grid(total)), dim3(NTHR), 0, stream, (const T*)dy,
                                  idx, (T*)dx, N, Ho, Wo, C / 8);)
    VC_LAUNCH_CHECK("vc_maxpool2d_bwd");
    return VC_OK;
  }
  const dim3 grid(ceil_div(W * (C / 8), NTHR), H, N);
  DISPATCH_T(dtype, "vc_maxpool2d_bwd",
             hipLaunchKernelGGL((maxpool_bwd_kernel<T>), grid, dim3(NTHR), 0, stream, (const T*)dy, idx, (T*)dx, N, H, W,
                                C / 8, Ho, Wo, k, s, p);)
  VC_LAUNCH_CHECK("vc_maxpool2d_bwd");
  return VC_OK;
}

int vc_avgpool_fwd(int dtype, const void* x, void* y, int N, int S, int C, hipStream_t stream) {
  VC_CHECK_ARG(x && y, "vc_avgpool_fwd: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0 && N > 0 && S > 0, "vc_avgpool_fwd: bad shape");
  if (S >= 256) {
    const int64_t pairs = (int64_t)N * (C / 8);
    DISPATCH_T(dtype, "vc_avgpool_fwd",
               hipLaunchKernelGGL((avgpool_fwd_wave_kernel<T>), dim3((unsigned)ceil_div64(pairs, NTHR / 64)), dim3(NTHR),
                                  0, stream, (const T*)x, (T*)y, N, S, C / 8);)
  } else {
    DISPATCH_T(dtype, "vc_avgpool_fwd",
               hipLaunchKernelGGL((avgpool_fwd_kernel<T>), dim3(ew_grid((int64_t)N * (C / 8))), dim3(NTHR), 0, stream,
                                  (const T*)x, (T*)y, N, S, C / 8);)
  }
  VC_LAUNCH_CHECK("vc_avgpool_fwd");
  return VC_OK;
}

int vc_avgpool_bwd(int dtype, const void* dy, void* dx, int N, int S, int C, hipStream_t stream) {
  VC_CHECK_ARG(dy && dx, "vc_avgpool_bwd: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0 && N > 0 && S > 0, "vc_avgpool_bwd: bad shape");
  DISPATCH_T(dtype, "vc_avgpool_bwd",
             hipLaunchKernelGGL((avgpool_bwd_kernel<T>), dim3(ew_grid((int64_t)N * S * (C / 8))), dim3(NTHR), 0,
                                stream, (const T*)dy, (T*)dx, N, S, C / 8);)
  VC_LAUNCH_CHECK("vc_avgpool_bwd");
  return VC_OK;
}

int vc_elementwise(int dtype, int op, const void* a, const void* b, void* out, int64_t n, hipStream_t stream) {
  VC_CHECK_ARG(a && out, "vc_elementwise: null pointer");
  VC_CHECK_SHAPE(n % 8 == 0 && n > 0, "vc_elementwise: element count must be a positive multiple of 8");
  VC_CHECK_ARG(op >= 0 && op <= 5, "vc_elementwise: unknown op %d", op);
  VC_CHECK_ARG(!(op == 0 || op == 1 || op == 4) || b, "vc_elementwise: op %d needs two inputs", op);
  const int64_t n8 = n / 8;
#define EW_CASE(OPN) \
  case OPN:          \
    hipLaunchKernelGGL((ew_kernel<T, OPN>), dim3(ew_grid(n8)), dim3(NTHR), 0, stream, (const T*)a, (const T*)b, (T*)out, n8); \
    break;
  DISPATCH_T(dtype, "vc_elementwise", switch (op) { EW_CASE(0) EW_CASE(1) EW_CASE(2) EW_CASE(3) EW_CASE(4) EW_CASE(5) })
#undef EW_CASE
  VC_LAUNCH_CHECK("vc_elementwise");
  return VC_OK;
}

int vc_cast(int src_dtype, int dst_dtype, const void* a, void* out, int64_t n, hipStream_t stream) {
  VC_CHECK_ARG(a && out && n > 0, "vc_cast: bad argument");
  dim3 grid(ew_grid(n)), blk(NTHR);
  if (src_dtype == VC_F32 && dst_dtype == VC_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, blk, 0, stream, (const float*)a, (bf16*)out, n);
  else if (src_dtype == VC_BF16 && dst_dtype == VC_F32)
    hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, blk, 0, stream, (const bf16*)a, (float*)out, n);
  else if (src_dtype == VC_F32 && dst_dtype == VC_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), grid, blk, 0, stream, (const float*)a, (float*)out, n);
  else if (src_dtype == VC_BF16 && dst_dtype == VC_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16, bf16>), grid, blk, 0, stream, (const bf16*)a, (bf16*)out, n);
  else { vc_set_error("vc_cast: unsupported dtypes %d -> %d", src_dtype, dst_dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_cast");
  return VC_OK;
}

// out[c] += sum over rows (caller zeroes or keeps a running gradient)
int vc_colsum(int dtype, const void* a, float* out, int M, int C, int lda, hipStream_t stream) {
  VC_CHECK_ARG(a && out && M > 0 && C > 0 && lda >= C, "vc_colsum: bad argument");
  const bool vec = lda % 8 == 0;
  // a block's threads walk their rows one load after the other, so short matrices (the LSTM's 512 x 2048 gate
  // gradients) are cut into many blocks of few rows: 256 rows per block took 53 us there
  int rows_per_block = vec ? 16 : 64;
  while (ceil_div(M, rows_per_block) > (vec ? 1024 : 512)) rows_per_block *= 2;
  const int slots = ceil_div(M, rows_per_block);
  // deterministic mode: a copy of out per block of rows, added in block order afterwards
  const bool ordered = vc_deterministic() && slots > 1;
  float* dst = out;
  if (ordered) {
    dst = vc_det_workspace((size_t)slots * C, stream, "vc_colsum");
    if (!dst) return VC_ERR_ARG;
  }
  const size_t slot_stride = ordered ? (size_t)C : 0;
  if (vec) {
    // rows are dense with `lda` (a multiple of 8) channels, of which the first C are summed
    DISPATCH_T(dtype, "vc_colsum",
               hipLaunchKernelGGL((colsum_vec_kernel<T>), dim3(slots), dim3(NTHR), 0, stream, (const T*)a, dst, M, lda,
                                  C, rows_per_block, slot_stride);)
  } else {
    dim3 grid(ceil_div(C, 64), slots);
    DISPATCH_T(dtype, "vc_colsum",
               hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(64), 0, stream, (const T*)a, dst, M, C, lda,
                                  rows_per_block, slot_stride);)
  }
  VC_LAUNCH_CHECK("vc_colsum");
  if (ordered) return vc_det_reduce(out, dst, (size_t)C, slots, stream);
  return VC_OK;
}

int vc_lstm_cell_fwd(int dtype, const float* gates_x, int ldx, const float* gates_h, const float* c_prev, void* h_out,
                     int ldh, float* c_out, float* act_out, int B, int H, hipStream_t stream) {
  VC_CHECK_ARG(gates_x && h_out && c_out && act_out, "vc_lstm_cell_fwd: null pointer");
  DISPATCH_T(dtype, "vc_lstm_cell_fwd",
             hipLaunchKernelGGL((lstm_cell_fwd_kernel<T>), dim3(ceil_div(B * H, NTHR)), dim3(NTHR), 0, stream,
                                gates_x, ldx, gates_h, c_prev, (T*)h_out, ldh, c_out, act_out, B, H);)
  VC_LAUNCH_CHECK("vc_lstm_cell_fwd");
  return VC_OK;
}

int vc_lstm_cell_bwd(int dtype, const void* dh_out, int ldo, const float* dh_rec, const float* dc_next,
                     const float* act, const float* c, const float* c_prev, void* dgates, float* dc_prev, int B, int H,
                     hipStream_t stream) {
  VC_CHECK_ARG(act && c && dgates && dc_prev, "vc_lstm_cell_bwd: null pointer");
  DISPATCH_T(dtype, "vc_lstm_cell_bwd",
             hipLaunchKernelGGL((lstm_cell_bwd_kernel<T>), dim3(ceil_div(B * H, NTHR)), dim3(NTHR), 0, stream,
                                (const T*)dh_out, ldo, dh_rec, dc_next, act, c, c_prev, (T*)dgates, dc_prev, B, H);)
  VC_LAUNCH_CHECK("vc_lstm_cell_bwd");
  return VC_OK;
}

// loss[0] and correct[0] are accumulated into (caller zeroes them); dlogits has `ldd` columns,
// the ones beyond C are written as zero.
int vc_ce_fwd_bwd(int grad_dtype, const float* logits, int ld, const int64_t* target, float* loss, int64_t* pred,
                  int* correct, void* dlogits, int ldd, int B, int C, float loss_scale, float grad_scale,
                  hipStream_t stream) {
  VC_CHECK_ARG(logits && target && loss, "vc_ce_fwd_bwd: null pointer");
  VC_CHECK_SHAPE(B > 0 && C > 0 && ld >= C && (!dlogits || ldd >= C), "vc_ce_fwd_bwd: bad shape");
  dim3 grid(ceil_div(B, 4)), blk(256);
  // deterministic mode: the B terms of the mean are added in row order (the count of correct rows is an integer sum)
  const bool ordered = vc_deterministic() && B > 1;
  float* dst = loss;
  if (ordered) {
    dst = vc_det_workspace((size_t)B, stream, "vc_ce_fwd_bwd");
    if (!dst) return VC_ERR_ARG;
  }
  if (grad_dtype == VC_BF16)
    hipLaunchKernelGGL((ce_kernel<bf16>), grid, blk, 0, stream, logits, ld, target, dst, pred, correct,
                       (bf16*)dlogits, ldd, B, C, loss_scale, grad_scale, (int)ordered);
  else if (grad_dtype == VC_F32)
    hipLaunchKernelGGL((ce_kernel<float>), grid, blk, 0, stream, logits, ld, target, dst, pred, correct,
                       (float*)dlogits, ldd, B, C, loss_scale, grad_scale, (int)ordered);
  else { vc_set_error("vc_ce_fwd_bwd: unsupported dtype %d", grad_dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_ce_fwd_bwd");
  if (ordered) return vc_det_reduce(loss, dst, 1, B, stream);  // one "table entry", B slots
  return VC_OK;
}

// lr_dev / step_dev (nullable) override lr / step with values read from device memory at execution time
int vc_optim_step(int mode, float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                  float eps, float wd, int step, float grad_scale, const float* lr_dev, const int64_t* step_dev,
                  hipStream_t stream) {
  VC_CHECK_ARG(p && g && m && n > 0 && (step >= 1 || step_dev), "vc_optim_step: bad argument");
  VC_CHECK_ARG(mode == 2 || v, "vc_optim_step: second-moment buffer missing");
  VC_CHECK_ARG(mode >= 0 && mode <= 2, "vc_optim_step: unknown mode %d", mode);
  hipLaunchKernelGGL(optim_kernel, dim3(ew_grid(n)), dim3(NTHR), 0, stream, p, g, m, v, n, mode, lr, b1, b2, eps, wd,
                     step, grad_scale, lr_dev, step_dev);
  VC_LAUNCH_CHECK("vc_optim_step");
  return VC_OK;
}

// *counter += inc on the stream (device-resident step counter of the optimizer / dropout generator)
int vc_counter_add(int64_t* counter, int64_t inc, hipStream_t stream) {
  VC_CHECK_ARG(counter, "vc_counter_add: null pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, stream, counter, inc);
  VC_LAUNCH_CHECK("vc_counter_add");
  return VC_OK;
}

}  // extern "C"
