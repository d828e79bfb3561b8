// Shared device/host helpers for the gfx950 (MI355X) kernels of libvc_hip.
// Everything here is header-only; kernels live in the vc_*.hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// ---------------------------------------------------------------------------
// Element types.  dtype codes are part of the C ABI (include/vc_hip.h).
// ---------------------------------------------------------------------------
#define VC_F32 0
#define VC_BF16 1

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Status codes / error string (returned through the C ABI, never thrown).
// ---------------------------------------------------------------------------
#define VC_OK 0
#define VC_ERR_SHAPE (-1)
#define VC_ERR_DTYPE (-2)
#define VC_ERR_HIP (-3)
#define VC_ERR_ARG (-4)

void vc_set_error(const char* fmt, ...);
constexpr int VC_MAX_DEVICES = 64;  // per-device tables of the library (workspaces, scratch addresses, CU counts)
int vc_device_cus();  // compute units of the current device (256 if no device answers)
// Deterministic mode (vc_set_deterministic): sums that blocks otherwise combine with fp32 atomics, in whatever order
// they finish, go through a workspace instead -- block (slot) s adds into ws[s * n + i], and vc_det_reduce then does
// out[i] += ws[0 * n + i] + ws[1 * n + i] + ... in slot order.
bool vc_deterministic();
size_t vc_det_capacity();  // floats of the current device's workspace (0: none registered)
// the current device's workspace with its first `floats` floats zeroed on `stream`; nullptr (and the error set) when
// it is missing or too small
float* vc_det_workspace(size_t floats, hipStream_t stream, const char* who);
int vc_det_reduce(float* out, const float* ws, size_t n, int slots, hipStream_t stream);

#define VC_CHECK_ARG(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      vc_set_error(__VA_ARGS__);       \
      return VC_ERR_ARG;               \
    }                                  \
  } while (0)

#define VC_CHECK_SHAPE(cond, ...)      \
  do {                                 \
    if (!(cond)) {                     \
      vc_set_error(__VA_ARGS__);       \
      return VC_ERR_SHAPE;             \
    }                                  \
  } while (0)

#define VC_LAUNCH_CHECK(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      vc_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return VC_ERR_HIP;                                                   \
    }                                                                      \
  } while (0)

// ---------------------------------------------------------------------------
// Division by a runtime-constant positive divisor with one mul-hi + shift.
// Exact for 0 <= n < 2^31 (all tensor indices here are below that).
// ---------------------------------------------------------------------------
struct FastDiv {
  uint32_t mul;
  uint32_t shr;
  uint32_t d;
};

static inline FastDiv make_fastdiv(int d) {
  FastDiv f;
  f.d = (uint32_t)d;
  if (d <= 1) {
    f.mul = 0;
    f.shr = 0;
    f.d = 1;
    return f;
  }
  int lg = 0;
  while ((1u << lg) < (uint32_t)d) ++lg;  // ceil(log2 d)
  int p = 31 + lg;
  f.mul = (uint32_t)(((1ull << p) + (uint64_t)d - 1) / (uint64_t)d);
  f.shr = (uint32_t)(p - 32);
  return f;
}

__device__ __forceinline__ int fd_div(int n, const FastDiv& f) {
  return f.d == 1 ? n : (int)(__umulhi((uint32_t)n, f.mul) >> f.shr);
}
__device__ __forceinline__ void fd_divmod(int n, const FastDiv& f, int& q, int& r) {
  q = fd_div(n, f);
  r = n - q * (int)f.d;
}

// ---------------------------------------------------------------------------
// Scalar conversions.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 8-element vectors of T as 8 floats (the unit every channel-vectorised kernel uses).
struct F8 {
  float v[8];
};

__device__ __forceinline__ F8 load8(const float* p) {
  F8 r;
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[i] = a[i];
    r.v[4 + i] = b[i];
  }
  return r;
}
__device__ __forceinline__ F8 load8(const bf16* p) {
  F8 r;
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (float)a[i];
  return r;
}
__device__ __forceinline__ void store8(float* p, const F8& r) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = r.v[i];
    b[i] = r.v[4 + i];
  }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16* p, const F8& r) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16)r.v[i];
  *reinterpret_cast<bf16x8*>(p) = a;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

#if defined(__HIPCC__)
// Exact (erf) GELU pieces without libm's erff: Phi(x) and phi(x) of the standard normal distribution from ONE
// exponential, e = exp(-x^2 / 2).  erfc(z) = (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z = |x| / sqrt(2)
// (Abramowitz & Stegun 7.1.26, absolute error <= 1.5e-7), so Phi(-|x|) = erfc(z) / 2 keeps its relative accuracy in
// the tail and Phi(|x|) = 1 - Phi(-|x|).  Measured against double precision over [-12, 12] in fp32 arithmetic:
// Phi 3.0e-7, x Phi(x) 4.2e-7, Phi + x phi 3.2e-7 absolute -- the same as the erff form (4.5e-7 on x Phi(x)) at a
// third of its instructions.  The GEMM epilogues that apply GELU or its derivative to every output element are bound
// by exactly this arithmetic.
__device__ __forceinline__ void vc_normal_cdf_pdf(float x, float& cdf, float& pdf) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  const float e = __expf(-0.5f * x * x);
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float q = 0.5f * poly * t * e;
  cdf = x < 0.f ? q : 1.f - q;
  pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float vc_gelu(float x) {  // x * Phi(x): nn.GELU() (approximate='none')
  float cdf, pdf;
  vc_normal_cdf_pdf(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float vc_gelu_grad(float x) {  // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
  float cdf, pdf;
  vc_normal_cdf_pdf(x, cdf, pdf);
  return fmaf(x, pdf, cdf);
}
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Geometry of one convolution (or linear layer: all spatial extents 1) as the
// implicit-GEMM kernels see it.  "src" is the tensor the im2col gather reads.
struct ConvGeom {
  int N, Di, Hi, Wi, Ci;  // gather-source activation, NDHWC
  int Do, Ho, Wo;         // pixel space that indexes GEMM rows
  int kd, kh, kw;
  int sd, sh, sw;
  int pd, ph, pw;
  int Co;   // GEMM columns (output channels of this GEMM)
  int K;    // kd*kh*kw*Ci
  int M;    // N*Do*Ho*Wo
  int ldc;  // row stride of the output matrix, in elements
  int ldi;  // row (pixel) stride of the gather source, in elements (>= Ci)
  FastDiv fWo, fHo, fDo;  // pixel index -> (n, od, oh, ow)
  FastDiv fC8, fkw, fkh;  // k-chunk index -> (tap, c) -> (rd, rh, rw)
  // Phase view (data gradient of a strided convolution, one GEMM per output phase): the GEMM's taps
  // (ah, aw) are every stride-th tap of the packed weight and its rows (hc, wc) every stride-th output pixel.
  int ldw;                         // row stride of the packed weight, in elements
  int wt_kh, wt_kw;                // kernel extent of the packed weight
  int wt_h0, wt_hs, wt_w0, wt_ws;  // weight tap rh = wt_h0 + wt_hs*ah, rw = wt_w0 + wt_ws*aw
  int oHo, oWo;                    // extent of the image the output rows live in
  int o_h0, o_hs, o_w0, o_ws;      // output pixel h = o_h0 + o_hs*hc, w = o_w0 + o_ws*wc
  // Split reduction of a single-tap layer (persistent NT kernel): work unit v = tile * ksplit + slice; a slice
  // covers 1 / ksplit of the channels and writes its own fp32 partial [slice][M][ldc].  1 = whole reduction.
  int ksplit;
};
