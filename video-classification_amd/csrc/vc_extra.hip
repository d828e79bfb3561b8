// Dropout and the TCG-LSTM fusion kernels (gfx950).  All bandwidth-bound and tiny next to the
// encoders; they exist so that no arithmetic of the path runs outside libvc_hip.
#include "vc_common.h"

namespace {

constexpr int NTHR = 256;

inline int ew_grid(int64_t n) {
  int64_t b = ceil_div64(n, NTHR);
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// counter-based generator: one 64-bit mix of (seed, element index) per element
__device__ __forceinline__ uint32_t mix_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}

template <typename T>
__global__ void dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, uint8_t* __restrict__ mask,
                                   int64_t n, uint32_t thresh, float scale, uint64_t seed,
                                   const int64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += (uint64_t)*seed_dev * 0xD1B54A32D192ED03ull;  // advances with the device step counter
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool keep = mix_u32(seed, (uint64_t)i) >= thresh;
    mask[i] = keep;
    out[i] = from_f32<T>(keep ? to_f32(x[i]) * scale : 0.f);
  }
}

template <typename T>
__global__ void dropout_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ mask, T* __restrict__ dx,
                                   int64_t n, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = from_f32<T>(mask[i] ? to_f32(dy[i]) * scale : 0.f);
}

// stochastic depth: the whole residual branch of a sample is kept (scaled by 1/(1-p)) or zeroed; the keep
// decision is a pure function of (seed, sample), so backward re-applies the same kernel to the gradient
template <typename T>
__global__ void drop_path_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t n8, int64_t chunks_per_sample,
                                 uint32_t thresh, float scale, uint64_t seed, const int64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += (uint64_t)*seed_dev * 0xD1B54A32D192ED03ull;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const bool keep = mix_u32(seed, (uint64_t)(i / chunks_per_sample)) >= thresh;
    F8 v = load8(x + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] = keep ? v.v[j] * scale : 0.f;
    store8(out + i * 8, v);
  }
}

// the per-sample factors drop_path_kernel applies, as a vector (for a GEMM epilogue that scales its rows)
__global__ void drop_path_scales_kernel(float* __restrict__ scales, int samples, uint32_t thresh, float scale,
                                        uint64_t seed, const int64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += (uint64_t)*seed_dev * 0xD1B54A32D192ED03ull;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < samples) scales[i] = mix_u32(seed, (uint64_t)i) >= thresh ? scale : 0.f;
}

// ---------------------------------------------------------------------------
// TCG-LSTM fusion.  Layouts are time-major: h_s [T][B][2H], h_l [Tl][B][H].
// Linear interpolation follows F.interpolate(mode='linear', align_corners=False):
//   src = max((t + 0.5) * Tl / T - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, Tl - 1).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void interp_coef(int t, int T, int Tl, int& i0, int& i1, float& lam) {
  float src = ((float)t + 0.5f) * ((float)Tl / (float)T) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > Tl - 1) i0 = Tl - 1;
  i1 = i0 + 1 < Tl ? i0 + 1 : Tl - 1;
  lam = src - (float)i0;
}

// cat[t][b] = [h_s (2H) | up (H) | up (H)]  in T
template <typename T>
__global__ void tcg_prep_kernel(const T* __restrict__ hs, const T* __restrict__ hl, T* __restrict__ cat, int Tn, int Tl,
                                int B, int H) {
  const int64_t total = (int64_t)Tn * B * 4 * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % (4 * H));
    const int64_t tb = i / (4 * H);
    const int b = (int)(tb % B), t = (int)(tb / B);
    float v;
    if (f < 2 * H) {
      v = to_f32(hs[((size_t)t * B + b) * 2 * H + f]);
    } else {
      const int j = (f - 2 * H) % H;
      int i0, i1;
      float lam;
      interp_coef(t, Tn, Tl, i0, i1, lam);
      v = (1.f - lam) * to_f32(hl[((size_t)i0 * B + b) * H + j]) + lam * to_f32(hl[((size_t)i1 * B + b) * H + j]);
    }
    cat[i] = from_f32<T>(v);
  }
}

// One workgroup per batch element.  z [T][B][2H] fp32 = gate pre-activation.
//   gate = sigmoid(z); h = gate*h_s + (1-gate)*up2; s_t = mean_f h; alpha = softmax_t(s); clip = sum_t alpha_t h_t
// Saves gate (fp32 [T][B][2H]), h (fp32 [T][B][2H]) and alpha (fp32 [B][T]) for backward.
template <typename T>
__global__ void tcg_fuse_fwd_kernel(const float* __restrict__ z, const T* __restrict__ cat, float* __restrict__ gate,
                                    float* __restrict__ h, float* __restrict__ alpha, T* __restrict__ clip, int Tn,
                                    int B, int H) {
  __shared__ float s_t[64];
  __shared__ float red[NTHR];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int F = 2 * H;
  for (int t = 0; t < Tn; ++t) {
    float part = 0.f;
    for (int f = tid; f < F; f += NTHR) {
      const size_t o = ((size_t)t * B + b) * F + f;
      const float g = 1.f / (1.f + __expf(-z[o]));
      const float a = to_f32(cat[((size_t)t * B + b) * 4 * H + f]);
      const float u = to_f32(cat[((size_t)t * B + b) * 4 * H + 2 * H + f]);
      const float hv = g * a + (1.f - g) * u;
      gate[o] = g;
      h[o] = hv;
      part += hv;
    }
    red[tid] = part;
    __syncthreads();
    for (int sft = NTHR / 2; sft > 0; sft >>= 1) {
      if (tid < sft) red[tid] += red[tid + sft];
      __syncthreads();
    }
    if (tid == 0) s_t[t] = red[0] / (float)F;
    __syncthreads();
  }
  if (tid == 0) {
    float mx = -INFINITY;
    for (int t = 0; t < Tn; ++t) mx = fmaxf(mx, s_t[t]);
    float se = 0.f;
    for (int t = 0; t < Tn; ++t) se += __expf(s_t[t] - mx);
    for (int t = 0; t < Tn; ++t) {
      const float a = __expf(s_t[t] - mx) / se;
      s_t[t] = a;
      alpha[(size_t)b * Tn + t] = a;
    }
  }
  __syncthreads();
  for (int f = tid; f < F; f += NTHR) {
    float acc = 0.f;
    for (int t = 0; t < Tn; ++t) acc = fmaf(s_t[t], h[((size_t)t * B + b) * F + f], acc);
    clip[(size_t)b * F + f] = from_f32<T>(acc);
  }
}

// Backward of the fusion: from dclip [B][2H] to dz [T][B][2H] (T, feeds the gate GEMM's backward),
// dhs [T][B][2H] (fp32) and dup2 [T][B][2H] (fp32; the caller adds the gate GEMM's data gradient).
template <typename T>
__global__ void tcg_fuse_bwd_kernel(const T* __restrict__ dclip, const T* __restrict__ cat,
                                    const float* __restrict__ gate, const float* __restrict__ h,
                                    const float* __restrict__ alpha, T* __restrict__ dz, float* __restrict__ dhs,
                                    float* __restrict__ dup2, int Tn, int B, int H) {
  __shared__ float da[64];
  __shared__ float red[NTHR];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int F = 2 * H;
  // dalpha_t = <dclip, h_t>
  for (int t = 0; t < Tn; ++t) {
    float part = 0.f;
    for (int f = tid; f < F; f += NTHR) part = fmaf(to_f32(dclip[(size_t)b * F + f]), h[((size_t)t * B + b) * F + f], part);
    red[tid] = part;
    __syncthreads();
    for (int sft = NTHR / 2; sft > 0; sft >>= 1) {
      if (tid < sft) red[tid] += red[tid + sft];
      __syncthreads();
    }
    if (tid == 0) da[t] = red[0];
    __syncthreads();
  }
  if (tid == 0) {
    float dot = 0.f;
    for (int t = 0; t < Tn; ++t) dot = fmaf(alpha[(size_t)b * Tn + t], da[t], dot);
    // ds_t = alpha_t (dalpha_t - dot); s_t is a mean over F features
    for (int t = 0; t < Tn; ++t) da[t] = alpha[(size_t)b * Tn + t] * (da[t] - dot) / (float)F;
  }
  __syncthreads();
  for (int t = 0; t < Tn; ++t) {
    const float al = alpha[(size_t)b * Tn + t], ds = da[t];
    for (int f = tid; f < F; f += NTHR) {
      const size_t o = ((size_t)t * B + b) * F + f;
      const float dh = al * to_f32(dclip[(size_t)b * F + f]) + ds;
      const float g = gate[o];
      const float a = to_f32(cat[((size_t)t * B + b) * 4 * H + f]);
      const float u = to_f32(cat[((size_t)t * B + b) * 4 * H + 2 * H + f]);
      dz[o] = from_f32<T>(dh * (a - u) * g * (1.f - g));
      dhs[o] = dh * g;
      dup2[o] = dh * (1.f - g);
    }
  }
}

// Combine the gate GEMM's data gradient dcat [T][B][4H] (T) with the direct paths:
//   dh_s = dhs + dcat[:, :2H];  dup = (dup2 + dcat[:, 2H:]) folded over the two copies;  dh_l = interp^T(dup)
// One thread per (tl, b, j) gathers from the <= T steps that read it (no atomics, deterministic).
template <typename T>
__global__ void tcg_combine_bwd_kernel(const T* __restrict__ dcat, const float* __restrict__ dhs,
                                       const float* __restrict__ dup2, T* __restrict__ dhs_out, T* __restrict__ dhl,
                                       int Tn, int Tl, int B, int H) {
  const int64_t n1 = (int64_t)Tn * B * 2 * H;
  const int64_t n2 = (int64_t)Tl * B * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int f = (int)(i % (2 * H));
      const int64_t tb = i / (2 * H);
      dhs_out[i] = from_f32<T>(dhs[i] + to_f32(dcat[tb * 4 * H + f]));
    } else {
      const int64_t k = i - n1;
      const int j = (int)(k % H);
      const int b = (int)((k / H) % B);
      const int tl = (int)(k / ((int64_t)H * B));
      float acc = 0.f;
      for (int t = 0; t < Tn; ++t) {
        int i0, i1;
        float lam;
        interp_coef(t, Tn, Tl, i0, i1, lam);
        float w = 0.f;
        if (i0 == tl) w += 1.f - lam;
        if (i1 == tl) w += lam;
        if (w != 0.f) {
          const size_t o = ((size_t)t * B + b);
          const float d = dup2[o * 2 * H + j] + dup2[o * 2 * H + H + j] + to_f32(dcat[o * 4 * H + 2 * H + j]) +
                          to_f32(dcat[o * 4 * H + 3 * H + j]);
          acc = fmaf(w, d, acc);
        }
      }
      dhl[k] = from_f32<T>(acc);
    }
  }
}

// short_in[t][b] = [g | mean_k c]  (time-major out, batch-major fp32 in: g [B][T][E], c [B][T][K][E]);
// long_in[tl][b] = g[b][tl*down]
template <typename T>
__global__ void tcg_inputs_kernel(const float* __restrict__ g, const float* __restrict__ c, T* __restrict__ short_in,
                                  T* __restrict__ long_in, int B, int Tn, int Tl, int K, int E, int down) {
  const int64_t n1 = (int64_t)Tn * B * 2 * E, n2 = (int64_t)Tl * B * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int f = (int)(i % (2 * E));
      const int b = (int)((i / (2 * E)) % B);
      const int t = (int)(i / ((int64_t)2 * E * B));
      float v;
      if (f < E) {
        v = g[((size_t)b * Tn + t) * E + f];
      } else {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += c[(((size_t)b * Tn + t) * K + k) * E + (f - E)];
        v = acc / (float)K;
      }
      short_in[i] = from_f32<T>(v);
    } else {
      const int64_t k2 = i - n1;
      const int f = (int)(k2 % E);
      const int b = (int)((k2 / E) % B);
      const int tl = (int)(k2 / ((int64_t)E * B));
      long_in[k2] = from_f32<T>(g[((size_t)b * Tn + tl * down) * E + f]);
    }
  }
}

// gradients of tcg_inputs: dg [B][T][E] fp32, dc [B][T][K][E] fp32
template <typename T>
__global__ void tcg_inputs_bwd_kernel(const T* __restrict__ dshort, const T* __restrict__ dlong, float* __restrict__ dg,
                                      float* __restrict__ dc, int B, int Tn, int Tl, int K, int E, int down) {
  const int64_t n = (int64_t)B * Tn * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % E);
    const int t = (int)((i / E) % Tn);
    const int b = (int)(i / ((int64_t)E * Tn));
    const size_t so = ((size_t)t * B + b) * 2 * E;
    float v = to_f32(dshort[so + f]);
    if (t % down == 0 && t / down < Tl) v += to_f32(dlong[((size_t)(t / down) * B + b) * E + f]);
    dg[i] = v;
    if (dc) {
      const float dm = to_f32(dshort[so + E + f]) / (float)K;
      for (int k = 0; k < K; ++k) dc[(((size_t)b * Tn + t) * K + k) * E + f] = dm;
    }
  }
}

}  // namespace

#define DISPATCH_T(dtype, who, ...)                              \
  if (dtype == VC_BF16) { typedef bf16 T; __VA_ARGS__ }          \
  else if (dtype == VC_F32) { typedef float T; __VA_ARGS__ }     \
  else { vc_set_error("%s: unsupported dtype %d", who, dtype); return VC_ERR_DTYPE; }

extern "C" {

// seed_dev (nullable): a device counter folded into the seed at execution time, so that a captured graph draws a
// new mask on every replay
int vc_dropout_fwd(int dtype, const void* x, void* out, uint8_t* mask, int64_t n, float p, uint64_t seed,
                   const int64_t* seed_dev, hipStream_t stream) {
  VC_CHECK_ARG(x && out && mask && n > 0 && p >= 0.f && p < 1.f, "vc_dropout_fwd: bad argument");
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  const float scale = 1.f / (1.f - p);
  DISPATCH_T(dtype, "vc_dropout_fwd",
             hipLaunchKernelGGL((dropout_fwd_kernel<T>), dim3(ew_grid(n)), dim3(NTHR), 0, stream, (const T*)x, (T*)out,
                                mask, n, thresh, scale, seed, seed_dev);)
  VC_LAUNCH_CHECK("vc_dropout_fwd");
  return VC_OK;
}

int vc_dropout_bwd(int dtype, const void* dy, const uint8_t* mask, void* dx, int64_t n, float p, hipStream_t stream) {
  VC_CHECK_ARG(dy && dx && mask && n > 0 && p >= 0.f && p < 1.f, "vc_dropout_bwd: bad argument");
  const float scale = 1.f / (1.f - p);
  DISPATCH_T(dtype, "vc_dropout_bwd",
             hipLaunchKernelGGL((dropout_bwd_kernel<T>), dim3(ew_grid(n)), dim3(NTHR), 0, stream, (const T*)dy, mask,
                                (T*)dx, n, scale);)
  VC_LAUNCH_CHECK("vc_dropout_bwd");
  return VC_OK;
}

int vc_drop_path(int dtype, const void* x, void* out, int64_t n, int64_t elems_per_sample, float p, uint64_t seed,
                 const int64_t* seed_dev, hipStream_t stream) {
  VC_CHECK_ARG(x && out && n > 0 && p >= 0.f && p < 1.f, "vc_drop_path: bad argument");
  VC_CHECK_SHAPE(n % 8 == 0 && elems_per_sample % 8 == 0 && elems_per_sample > 0 && n % elems_per_sample == 0,
                 "vc_drop_path: sizes must be multiples of 8 and of the per-sample size");
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  const float scale = 1.f / (1.f - p);
  DISPATCH_T(dtype, "vc_drop_path",
             hipLaunchKernelGGL((drop_path_kernel<T>), dim3(ew_grid(n / 8)), dim3(NTHR), 0, stream, (const T*)x,
                                (T*)out, n / 8, elems_per_sample / 8, thresh, scale, seed, seed_dev);)
  VC_LAUNCH_CHECK("vc_drop_path");
  return VC_OK;
}

int vc_drop_path_scales(float* scales, int samples, float p, uint64_t seed, const int64_t* seed_dev,
                        hipStream_t stream) {
  VC_CHECK_ARG(scales && samples > 0 && p >= 0.f && p < 1.f, "vc_drop_path_scales: bad argument");
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  hipLaunchKernelGGL(drop_path_scales_kernel, dim3((samples + 255) / 256), dim3(256), 0, stream, scales, samples,
                     thresh, 1.f / (1.f - p), seed, seed_dev);
  VC_LAUNCH_CHECK("vc_drop_path_scales");
  return VC_OK;
}

int vc_tcg_inputs(int dtype, const float* g, const float* c, void* short_in, void* long_in, int B, int T_, int Tl,
                  int K, int E, int down, hipStream_t stream) {
  VC_CHECK_ARG(g && c && short_in && long_in, "vc_tcg_inputs: null pointer");
  VC_CHECK_SHAPE(B > 0 && T_ > 0 && Tl == (T_ + down - 1) / down && K > 0 && E > 0, "vc_tcg_inputs: bad shape");
  const int64_t n = (int64_t)T_ * B * 2 * E + (int64_t)Tl * B * E;
  DISPATCH_T(dtype, "vc_tcg_inputs",
             hipLaunchKernelGGL((tcg_inputs_kernel<T>), dim3(ew_grid(n)), dim3(NTHR), 0, stream, g, c, (T*)short_in,
                                (T*)long_in, B, T_, Tl, K, E, down);)
  VC_LAUNCH_CHECK("vc_tcg_inputs");
  return VC_OK;
}

int vc_tcg_inputs_bwd(int dtype, const void* dshort, const void* dlong, float* dg, float* dc, int B, int T_, int Tl,
                      int K, int E, int down, hipStream_t stream) {
  VC_CHECK_ARG(dshort && dlong && dg, "vc_tcg_inputs_bwd: null pointer");
  const int64_t n = (int64_t)B * T_ * E;
  DISPATCH_T(dtype, "vc_tcg_inputs_bwd",
             hipLaunchKernelGGL((tcg_inputs_bwd_kernel<T>), dim3(ew_grid(n)), dim3(NTHR), 0, stream, (const T*)dshort,
                                (const T*)dlong, dg, dc, B, T_, Tl, K, E, down);)
  VC_LAUNCH_CHECK("vc_tcg_inputs_bwd");
  return VC_OK;
}

int vc_tcg_prep(int dtype, const void* hs, const void* hl, void* cat, int T_, int Tl, int B, int H,
                hipStream_t stream) {
  VC_CHECK_ARG(hs && hl && cat, "vc_tcg_prep: null pointer");
  const int64_t n = (int64_t)T_ * B * 4 * H;
  DISPATCH_T(dtype, "vc_tcg_prep",
             hipLaunchKernelGGL((tcg_prep_kernel<T>), dim3(ew_grid(n)), dim3(NTHR), 0, stream, (const T*)hs,
                                (const T*)hl, (T*)cat, T_, Tl, B, H);)
  VC_LAUNCH_CHECK("vc_tcg_prep");
  return VC_OK;
}

int vc_tcg_fuse_fwd(int dtype, const float* z, const void* cat, float* gate, float* h, float* alpha, void* clip,
                    int T_, int B, int H, hipStream_t stream) {
  VC_CHECK_ARG(z && cat && gate && h && alpha && clip, "vc_tcg_fuse_fwd: null pointer");
  VC_CHECK_SHAPE(T_ > 0 && T_ <= 64 && B > 0 && H > 0, "vc_tcg_fuse_fwd: T must be in 1..64");
  DISPATCH_T(dtype, "vc_tcg_fuse_fwd",
             hipLaunchKernelGGL((tcg_fuse_fwd_kernel<T>), dim3(B), dim3(NTHR), 0, stream, z, (const T*)cat, gate, h,
                                alpha, (T*)clip, T_, B, H);)
  VC_LAUNCH_CHECK("vc_tcg_fuse_fwd");
  return VC_OK;
}

int vc_tcg_fuse_bwd(int dtype, const void* dclip, const void* cat, const float* gate, const float* h,
                    const float* alpha, void* dz, float* dhs, float* dup2, int T_, int B, int H, hipStream_t stream) {
  VC_CHECK_ARG(dclip && cat && gate && h && alpha && dz && dhs && dup2, "vc_tcg_fuse_bwd: null pointer");
  VC_CHECK_SHAPE(T_ > 0 && T_ <= 64 && B > 0 && H > 0, "vc_tcg_fuse_bwd: T must be in 1..64");
  DISPATCH_T(dtype, "vc_tcg_fuse_bwd",
             hipLaunchKernelGGL((tcg_fuse_bwd_kernel<T>), dim3(B), dim3(NTHR), 0, stream, (const T*)dclip,
                                (const T*)cat, gate, h, alpha, (T*)dz, dhs, dup2, T_, B, H);)
  VC_LAUNCH_CHECK("vc_tcg_fuse_bwd");
  return VC_OK;
}

int vc_tcg_combine_bwd(int dtype, const void* dcat, const float* dhs, const float* dup2, void* dhs_out, void* dhl,
                       int T_, int Tl, int B, int H, hipStream_t stream) {
  VC_CHECK_ARG(dcat && dhs && dup2 && dhs_out && dhl, "vc_tcg_combine_bwd: null pointer");
  const int64_t n = (int64_t)T_ * B * 2 * H + (int64_t)Tl * B * H;
  DISPATCH_T(dtype, "vc_tcg_combine_bwd",
             hipLaunchKernelGGL((tcg_combine_bwd_kernel<T>), dim3(ew_grid(n)), dim3(NTHR), 0, stream, (const T*)dcat,
                                dhs, dup2, (T*)dhs_out, (T*)dhl, T_, Tl, B, H);)
  VC_LAUNCH_CHECK("vc_tcg_combine_bwd");
  return VC_OK;
}

}  // extern "C"
