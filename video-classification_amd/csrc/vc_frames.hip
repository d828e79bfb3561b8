// Per-frame input transform on the GPU: decoded uint8 frames [N, Hs, Ws, 3] -> normalised planar float / bf16
// frames [N, C, Hd, Wd], the chain the reference runs per frame on the host with PIL + torchvision
// (train.py:344-388; utils/common/data_loaders.py:39-51, 88-100):
//   [convert('L')] -> Resize (Pillow's anti-aliased bilinear, 8-bit fixed point) -> [horizontal flip]
//   -> ToTensor (/255) -> Normalize ((x - mean) / std).
// Integer-exact with Pillow: 22-bit fixed-point weights, int32 accumulation from 1 << 21, >> 22, clip to 0..255,
// horizontal pass first and its uint8 result feeding the vertical pass; the float tail uses correctly rounded
// IEEE division, so the fp32 output is bit-identical to the host chain as well.
//
// One workgroup produces TY output rows of one frame.  The source rows it needs are one contiguous byte range of
// the frame; they are copied to LDS with 16-byte loads, reduced horizontally into a uint8 LDS tile and reduced
// vertically from there, so HBM sees every source byte about once and every output element exactly once.
#include <math.h>

#include "vc_common.h"

// every floating-point step below has to round exactly as the host C code it restates: no fused multiply-add
#pragma clang fp contract(off)

#define VC_U8 2  // vc_frames_transform only: planar uint8 output, before ToTensor / Normalize

namespace {

constexpr int PREC = 22;  // 32 - 8 - 2, Pillow's PRECISION_BITS
constexpr int FT_THREADS = 512;
constexpr int FT_MAX_LDS = 150 * 1024;

struct FrameGeom {
  int n, hs, ws, hd, wd;
  int co;        // output channels: 3, or 1 when gray
  int gray;      // RGB -> L before the resize
  int ksx, ksy;  // row strides of the weight tables in global memory
  int kpx;       // ksx rounded up to a multiple of 4: taps are multiplied four at a time, the surplus with weight 0
  int wdp;       // wd rounded up to a multiple of 4: row stride of a plane of the horizontally reduced tile
  int ty;        // output rows per workgroup
  int cap_rows;  // most source rows a workgroup can need
  int raw_bytes, lum_bytes, tmp_bytes;  // LDS regions, each a multiple of 16 bytes
  FastDiv f_wd, f_w4, f_co;
};

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int byte_of(uint32_t d, int j) { return (int)((d >> (8 * j)) & 0xffu); }
// byte x weight: the weights of the triangle filter lie in 0 .. 2^22, so the 24-bit multiplier (full rate) is exact
__device__ __forceinline__ int mul24(int b, int w) { return (int)__umul24((unsigned)b, (unsigned)w); }

// One pixel of the horizontal pass: CS interleaved channels, taps x0 .. x0 + 4 * ng - 1 of one LDS row, weights k
// (zero beyond the window).  The bytes are fetched as aligned dwords and shifted into place.
template <int CS>
__device__ __forceinline__ void hpass_pixel(const uint8_t* rowbytes, int b0, const int* k, int ng, int (&acc)[CS]) {
  const uint32_t* d = reinterpret_cast<const uint32_t*>(rowbytes + (b0 & ~3));
  const uint32_t sh = (uint32_t)(b0 & 3);
#pragma unroll
  for (int c = 0; c < CS; ++c) acc[c] = 1 << (PREC - 1);
  uint32_t d0 = d[0];
  for (int gq = 0; gq < ng; ++gq) {
    const int4 w = *reinterpret_cast<const int4*>(k + 4 * gq);
    const int wv[4] = {w.x, w.y, w.z, w.w};
    uint32_t a[CS];
#pragma unroll
    for (int j = 0; j < CS; ++j) {
      const uint32_t d1 = d[CS * gq + j + 1];
      a[j] = __builtin_amdgcn_alignbyte(d1, d0, sh);
      d0 = d1;
    }
#pragma unroll
    for (int j = 0; j < 4 * CS; ++j) acc[j % CS] += mul24(byte_of(a[j >> 2], j & 3), wv[j / CS]);
  }
}

template <typename OUT>
__global__ __launch_bounds__(FT_THREADS) void frames_transform_kernel(FrameGeom g, const uint8_t* __restrict__ src,
                                                                      const int* __restrict__ xb,
                                                                      const int* __restrict__ xk,
                                                                      const int* __restrict__ yb,
                                                                      const int* __restrict__ yk,
                                                                      const uint8_t* __restrict__ flip,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ stdv,
                                                                      OUT* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* raw = lds;                                        // source rows (RGB), starting `mis` bytes in
  uint8_t* lum = raw + g.raw_bytes;                          // gray only: luma of the source rows
  uint8_t* tmp = lum + g.lum_bytes;                          // horizontally reduced rows [nrows][co][wdp]
  int* xkL = reinterpret_cast<int*>(tmp + g.tmp_bytes);      // [wd][kpx]
  int* xbL = xkL + g.wd * g.kpx;                             // [wdp] first source column of every window
  int* ykL = xbL + g.wdp;                                    // [ty][ksy]
  int* ybL = ykL + g.ty * g.ksy;                             // [ty] first tile row of every window
  float* lut = reinterpret_cast<float*>(ybL + ((g.ty + 3) & ~3));  // [co][256] normalised value of a byte

  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int y0 = blockIdx.x * g.ty;
  const int y1 = min(y0 + g.ty, g.hd);

  // source rows [r0, r0 + nrows) feed the output rows [y0, y1); everything read from the tables is clamped, so
  // that tables of another geometry give wrong pixels but never an access outside the frame or the tile
  int r0 = yb[2 * y0];
  int rend = yb[2 * (y1 - 1)] + yb[2 * (y1 - 1) + 1];
  r0 = max(0, min(r0, g.hs - 1));
  rend = max(r0 + 1, min(rend, g.hs));
  const int nrows = min(rend - r0, g.cap_rows);

  // --- tables ---
  for (int i = tid; i < g.wd * g.kpx; i += FT_THREADS) {
    const int x = i / g.kpx, t = i - x * g.kpx;
    xkL[i] = t < g.ksx ? xk[x * g.ksx + t] : 0;
  }
  for (int i = tid; i < g.wd; i += FT_THREADS) xbL[i] = max(0, min(xb[2 * i], g.ws - 1));
  for (int i = tid; i < (y1 - y0) * g.ksy; i += FT_THREADS) ykL[i] = yk[y0 * g.ksy + i];
  for (int i = tid; i < y1 - y0; i += FT_THREADS) ybL[i] = max(0, min(yb[2 * (y0 + i)] - r0, nrows - 1));
  if constexpr (sizeof(OUT) != 1) {
    for (int i = tid; i < g.co * 256; i += FT_THREADS) {
      const int c = i >> 8;
      const float f = (float)(i & 255) / 255.0f;  // ToTensor
      lut[i] = (f - mean[c]) / stdv[c];            // Normalize
    }
  }

  // --- stage the source rows: one contiguous byte range, 16-byte loads from a 16-byte aligned start ---
  const size_t row_bytes = (size_t)g.ws * 3;
  const uint8_t* gsrc = src + ((size_t)n * g.hs + r0) * row_bytes;
  const uint8_t* gend = src + (size_t)g.n * g.hs * row_bytes;
  const int mis = (int)(reinterpret_cast<uintptr_t>(gsrc) & 15);
  const uint8_t* gbase = gsrc - mis;  // >= src because src itself is 16-byte aligned
  const int nvec = (mis + nrows * (int)row_bytes + 15) >> 4;
  for (int i = tid; i < nvec; i += FT_THREADS) {
    const uint8_t* p = gbase + (size_t)i * 16;
    if (p + 16 <= gend) {
      *reinterpret_cast<uint4*>(raw + i * 16) = *reinterpret_cast<const uint4*>(p);
    } else {
      for (int b = 0; b < 16; ++b) raw[i * 16 + b] = (p + b < gend) ? p[b] : (uint8_t)0;
    }
  }
  __syncthreads();

  if (g.gray) {  // four pixels (12 source bytes) per step
    const int npix4 = (nrows * g.ws + 3) >> 2;
    const uint32_t* d = reinterpret_cast<const uint32_t*>(raw + (mis & ~3));
    const uint32_t sh = (uint32_t)(mis & 3);
    for (int i = tid; i < npix4; i += FT_THREADS) {
      const uint32_t d0 = d[3 * i], d1 = d[3 * i + 1], d2 = d[3 * i + 2], d3 = d[3 * i + 3];
      const uint32_t a[3] = {__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                             __builtin_amdgcn_alignbyte(d3, d2, sh)};
      uint32_t o = 0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = byte_of(a[(3 * p) >> 2], (3 * p) & 3), gg = byte_of(a[(3 * p + 1) >> 2], (3 * p + 1) & 3),
                  b = byte_of(a[(3 * p + 2) >> 2], (3 * p + 2) & 3);
        o |= (uint32_t)((r * 19595 + gg * 38470 + b * 7471 + 0x8000) >> 16) << (8 * p);
      }
      reinterpret_cast<uint32_t*>(lum)[i] = o;
    }
    __syncthreads();
  }

  // --- horizontal pass into tmp (planar per channel, so that the vertical pass reads four columns per dword) ---
  {
    const int ng = g.kpx >> 2;
    const int hitems = nrows * g.wd;
    for (int i = tid; i < hitems; i += FT_THREADS) {
      int r, x;
      fd_divmod(i, g.f_wd, r, x);
      const int x0 = xbL[x];
      const int* k = xkL + x * g.kpx;
      if (g.gray) {
        int acc[1];
        hpass_pixel<1>(lum, r * g.ws + x0, k, ng, acc);
        tmp[r * g.wdp + x] = (uint8_t)clip8(acc[0] >> PREC);
      } else {
        int acc[3];
        hpass_pixel<3>(raw, mis + (r * g.ws + x0) * 3, k, ng, acc);
#pragma unroll
        for (int c = 0; c < 3; ++c) tmp[(r * 3 + c) * g.wdp + x] = (uint8_t)clip8(acc[c] >> PREC);
      }
    }
  }
  __syncthreads();

  // --- vertical pass, ToTensor + Normalize through the table, flip; four columns per thread ---
  const bool fl = flip != nullptr && flip[n] != 0;
  const int w4 = g.wdp >> 2;
  const int vitems = (y1 - y0) * g.co * w4;
  const bool vec_ok = (g.wd & 3) == 0;
  for (int i = tid; i < vitems; i += FT_THREADS) {
    int t2, x4, yy, c;
    fd_divmod(i, g.f_w4, t2, x4);
    fd_divmod(t2, g.f_co, yy, c);
    const int q0 = ybL[yy];
    const int* k = ykL + yy * g.ksy;
    int acc[4] = {1 << (PREC - 1), 1 << (PREC - 1), 1 << (PREC - 1), 1 << (PREC - 1)};
    for (int t = 0; t < g.ksy; ++t) {
      const int row = min(q0 + t, nrows - 1);
      const uint32_t d = *reinterpret_cast<const uint32_t*>(tmp + (row * g.co + c) * g.wdp + 4 * x4);
      const int w = k[t];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += mul24(byte_of(d, j), w);
    }
    int bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = clip8(acc[j] >> PREC);
    OUT* orow = out + (((size_t)n * g.co + c) * g.hd + (y0 + yy)) * g.wd;
    const int x = 4 * x4;
    if constexpr (sizeof(OUT) == 1) {  // planar uint8: the resized (and flipped) frame itself
      if (vec_ok) {
        const uint32_t w = fl ? (uint32_t)(bv[3] | (bv[2] << 8) | (bv[1] << 16) | (bv[0] << 24))
                              : (uint32_t)(bv[0] | (bv[1] << 8) | (bv[2] << 16) | (bv[3] << 24));
        *reinterpret_cast<uint32_t*>(orow + (fl ? g.wd - 4 - x : x)) = w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x + j < g.wd) orow[fl ? g.wd - 1 - (x + j) : x + j] = (OUT)bv[j];
      }
    } else {
      float f[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = lut[c * 256 + bv[j]];
      if (vec_ok) {
        OUT v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = from_f32<OUT>(fl ? f[3 - j] : f[j]);
        OUT* dst = orow + (fl ? g.wd - 4 - x : x);
        if constexpr (sizeof(OUT) == 4) {
          *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(v);
        } else {
          *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(v);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x + j < g.wd) orow[fl ? g.wd - 1 - (x + j) : x + j] = from_f32<OUT>(f[j]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// ColorJitter on the resized uint8 frame (planar), then ToTensor + Normalize.  torchvision applies brightness,
// contrast, saturation and hue to a PIL image in an order drawn per call; each is Pillow byte arithmetic:
// Image.blend in float32 against black / the frame's rounded mean luma / the pixel's luma, and an RGB -> HSV -> RGB
// round trip with the uint8 hue shifted (wrap-around).  The float and double steps below follow Pillow's C code
// operation by operation, so the bytes are identical.
// ---------------------------------------------------------------------------------------------------------------
constexpr int JT_THREADS = 1024;

__device__ __forceinline__ int luma_of(int r, int g, int b) {
  return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16;
}

// Image.blend(im1, im2, alpha): (UINT8)(in1 + alpha * (in2 - in1)) inside [0, 1], clipped outside
__device__ __forceinline__ int blend_byte(int in1, int in2, float alpha, bool inside) {
  const float s = (float)in1 + alpha * (float)(in2 - in1);
  if (inside) return (int)s;
  return s <= 0.f ? 0 : (s >= 255.f ? 255 : (int)s);
}

__device__ __forceinline__ void rgb_to_hsv(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) {
    uh = 0;
    us = 0;
    return;
  }
  const float cr = (float)(maxc - minc);
  const float s = cr / (float)maxc;
  const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
  float h;
  if (r == maxc) h = bc - gc;
  else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
  else h = (float)(4.0 + (double)gc - (double)rc);
  h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
  uh = clip8((int)((double)h * 255.0));
  us = clip8((int)((double)s * 255.0));
}

__device__ __forceinline__ int round_half_away(double z) { return (int)(z >= 0.0 ? floor(z + 0.5) : ceil(z - 0.5)); }

__device__ __forceinline__ void hsv_to_rgb(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) {
    r = g = b = v;
    return;
  }
  const double hh = (double)(float)h * 6.0 / 255.0;
  const int i = (int)floor(hh);
  const double f = (double)(float)(hh - (double)(float)i);
  const double fs = (double)(float)((double)(float)s / 255.0);
  const double vf = (double)(float)v;
  const int p = clip8(round_half_away(vf * (1.0 - fs)));
  const int q = clip8(round_half_away(vf * (1.0 - fs * f)));
  const int t = clip8(round_half_away(vf * (1.0 - fs * (1.0 - f))));
  switch (i % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

struct JitterParams {
  float fb, fc, fs;
  bool in_b, in_c, in_s;
  int shift;
};

// one operation of the chain on one pixel; `mean` is the frame's rounded mean luma (contrast only)
__device__ __forceinline__ void jitter_op(int op, const JitterParams& jp, int mean, int& r, int& g, int& b) {
  if (op == 0) {
    r = blend_byte(0, r, jp.fb, jp.in_b);
    g = blend_byte(0, g, jp.fb, jp.in_b);
    b = blend_byte(0, b, jp.fb, jp.in_b);
  } else if (op == 1) {
    r = blend_byte(mean, r, jp.fc, jp.in_c);
    g = blend_byte(mean, g, jp.fc, jp.in_c);
    b = blend_byte(mean, b, jp.fc, jp.in_c);
  } else if (op == 2) {
    const int l = luma_of(r, g, b);
    r = blend_byte(l, r, jp.fs, jp.in_s);
    g = blend_byte(l, g, jp.fs, jp.in_s);
    b = blend_byte(l, b, jp.fs, jp.in_s);
  } else if (op == 3) {
    int h, s, v;
    rgb_to_hsv(r, g, b, h, s, v);
    hsv_to_rgb((h + jp.shift) & 255, s, v, r, g, b);
  }
}

// One workgroup per frame.  Every thread owns the pixels tid, tid + 1024, ...; the operations ahead of the
// contrast step are applied in a first sweep that also sums the luma (contrast blends against the mean luma of the
// frame as it is at that point of the chain), the rest in a second sweep that ends in the normalised output.
template <typename OUT>
__global__ __launch_bounds__(JT_THREADS) void frames_jitter_kernel(uint8_t* __restrict__ planes, int hw,
                                                                   const uint8_t* __restrict__ order,
                                                                   const float* __restrict__ factors,
                                                                   const uint8_t* __restrict__ hue_shift,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ stdv,
                                                                   OUT* __restrict__ out) {
  __shared__ float lut[3 * 256];
  __shared__ int red[JT_THREADS / 64];
  const int tid = threadIdx.x, n = blockIdx.x;
  uint8_t* P = planes + (size_t)n * 3 * hw;
  OUT* O = out + (size_t)n * 3 * hw;

  int ops[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ops[k] = order[4 * n + k];
  JitterParams jp;
  jp.fb = factors[3 * n];
  jp.fc = factors[3 * n + 1];
  jp.fs = factors[3 * n + 2];
  jp.in_b = jp.fb >= 0.f && jp.fb <= 1.f;
  jp.in_c = jp.fc >= 0.f && jp.fc <= 1.f;
  jp.in_s = jp.fs >= 0.f && jp.fs <= 1.f;
  jp.shift = hue_shift[n];
  int cpos = 4;  // position of the (first) contrast step; a second one is ignored
#pragma unroll
  for (int k = 3; k >= 0; --k)
    if (ops[k] == 1) cpos = k;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k > cpos && ops[k] == 1) ops[k] = 255;

  for (int i = tid; i < 3 * 256; i += JT_THREADS) {
    const int c = i >> 8;
    const float f = (float)(i & 255) / 255.0f;
    lut[i] = (f - mean[c]) / stdv[c];
  }

  int mean_l = 0;
  if (cpos < 4) {
    int sum = 0;
    for (int i = tid; i < hw; i += JT_THREADS) {
      int r = P[i], g = P[hw + i], b = P[2 * hw + i];
      if (cpos > 0) {
        for (int k = 0; k < cpos; ++k) jitter_op(ops[k], jp, 0, r, g, b);
        P[i] = (uint8_t)r;
        P[hw + i] = (uint8_t)g;
        P[2 * hw + i] = (uint8_t)b;
      }
      sum += luma_of(r, g, b);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    long long tot = 0;
    for (int w = 0; w < JT_THREADS / 64; ++w) tot += red[w];
    // int(mean + 0.5) of ImageStat's mean = sum / count
    mean_l = (int)((2 * tot + hw) / (2 * (long long)hw));
  } else {
    __syncthreads();  // the table
  }

  const int k0 = cpos < 4 ? cpos : 0;
  for (int i = tid; i < hw; i += JT_THREADS) {
    int r = P[i], g = P[hw + i], b = P[2 * hw + i];
    for (int k = k0; k < 4; ++k) jitter_op(ops[k], jp, mean_l, r, g, b);
    O[i] = from_f32<OUT>(lut[r]);
    O[hw + i] = from_f32<OUT>(lut[256 + g]);
    O[2 * hw + i] = from_f32<OUT>(lut[512 + b]);
  }
}

// rows of the source a block of `ty` output rows can touch (window of the first row to window of the last)
int rows_cap(int hs, int hd, int ty) {
  const double scale = (double)hs / hd;
  const double support = scale < 1.0 ? 1.0 : scale;
  const int cap = (int)floor((ty - 1) * scale + 2.0 * support + 1.0) + 2;
  return cap < hs ? cap : hs;
}

}  // namespace

extern "C" {

int vc_resample_ksize(int in_size, int out_size) {
  if (in_size < 1 || out_size < 1) return VC_ERR_ARG;
  double scale = (double)in_size / out_size;
  if (scale < 1.0) scale = 1.0;
  return (int)ceil(scale) * 2 + 1;
}

// Host code only.  Same operations in the same order as Pillow's precompute_coeffs + normalize_coeffs_8bpc
// (triangle filter, full-extent box), in double precision.
int vc_resample_coeffs(int in_size, int out_size, int* bounds, int* kk) {
  VC_CHECK_ARG(in_size >= 1 && out_size >= 1 && bounds && kk, "vc_resample_coeffs: bad arguments");
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  const double ss = 1.0 / filterscale;
  double* w = new double[ksize];
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < ksize; ++x) w[x] = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      w[x] = a < 1.0 ? 1.0 - a : 0.0;
      ww += w[x];
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) w[x] /= ww;
    for (int x = 0; x < ksize; ++x) {
      const double v = w[x] * (double)(1 << PREC);
      kk[xx * ksize + x] = w[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  delete[] w;
  return ksize;
}

int vc_frames_transform(const void* frames, int n, int hs, int ws, int gray, int hd, int wd, const int* xbounds,
                        const int* xk, int ksx, const int* ybounds, const int* yk, int ksy, const void* flip,
                        const float* mean, const float* stdv, void* out, int out_dtype, hipStream_t stream) {
  VC_CHECK_SHAPE(n >= 0 && hs >= 1 && ws >= 1 && hd >= 1 && wd >= 1, "vc_frames_transform: bad extents");
  VC_CHECK_ARG(out_dtype == VC_F32 || out_dtype == VC_BF16 || out_dtype == VC_U8, "vc_frames_transform: dtype %d",
               out_dtype);
  VC_CHECK_ARG(ksx == vc_resample_ksize(ws, wd) && ksy == vc_resample_ksize(hs, hd),
               "vc_frames_transform: weight tables of another geometry (ksize %d/%d, expected %d/%d)", ksx, ksy,
               vc_resample_ksize(ws, wd), vc_resample_ksize(hs, hd));
  if (n == 0) return VC_OK;
  VC_CHECK_ARG(frames && xbounds && xk && ybounds && yk && out && (out_dtype == VC_U8 || (mean && stdv)),
               "vc_frames_transform: null pointer");
  VC_CHECK_ARG((reinterpret_cast<uintptr_t>(frames) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               "vc_frames_transform: frames and out must be 16-byte aligned");
  VC_CHECK_SHAPE(n <= 65535, "vc_frames_transform: at most 65535 frames per call (got %d)", n);
  VC_CHECK_SHAPE((int64_t)hs * ws * 3 < (1ll << 31) && (int64_t)hd * wd * 3 < (1ll << 31),
                 "vc_frames_transform: frame too large");

  FrameGeom g;
  memset(&g, 0, sizeof(g));
  g.n = n; g.hs = hs; g.ws = ws; g.hd = hd; g.wd = wd;
  g.gray = gray ? 1 : 0;
  g.co = gray ? 1 : 3;
  g.ksx = ksx; g.ksy = ksy;
  g.kpx = (ksx + 3) & ~3;
  g.wdp = (wd + 3) & ~3;
  int lds = 0;
  // the largest block of rows whose tiles leave room for several workgroups per CU; smaller blocks for wide or
  // strongly reduced frames, down to one row
  const int tys[] = {16, 8, 4, 2, 1};
  for (int pass = 0; pass < 2 && lds == 0; ++pass) {
    const int limit = pass == 0 ? 52 * 1024 : FT_MAX_LDS;
    for (int ty : tys) {
      const int cap = rows_cap(hs, hd, ty);
      // the passes read up to kpx taps and one dword beyond a window: slack behind the row data
      const int slack = ((g.kpx * 3 + 8 + 15) & ~15) + 16;
      const int raw = ((cap * ws * 3 + 15 + 15) & ~15) + slack;
      const int lum = gray ? ((cap * ws + 15) & ~15) + slack : 0;
      const int tmp = (cap * g.co * g.wdp + 15) & ~15;
      const int tables = 4 * (wd * g.kpx + g.wdp + ty * ksy + ((ty + 3) & ~3) + g.co * 256);
      if (raw + lum + tmp + tables <= limit) {
        g.ty = ty; g.cap_rows = cap; g.raw_bytes = raw; g.lum_bytes = lum; g.tmp_bytes = tmp;
        lds = raw + lum + tmp + tables;
        break;
      }
    }
  }
  VC_CHECK_SHAPE(lds > 0, "vc_frames_transform: %d x %d -> %d x %d needs more than %d bytes of LDS per row block",
                 hs, ws, hd, wd, FT_MAX_LDS);
  g.f_wd = make_fastdiv(wd);
  g.f_w4 = make_fastdiv(g.wdp >> 2);
  g.f_co = make_fastdiv(g.co);

  dim3 grid(ceil_div(hd, g.ty), n);
  const uint8_t* s = static_cast<const uint8_t*>(frames);
  const uint8_t* fl = static_cast<const uint8_t*>(flip);
  auto launch = [&](auto kern, auto* o) -> int {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
      vc_set_error("vc_frames_transform: cannot reserve %d bytes of LDS", lds);
      return VC_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, grid, dim3(FT_THREADS), lds, stream, g, s, xbounds, xk, ybounds, yk, fl, mean, stdv, o);
    return VC_OK;
  };
  int rc;
  if (out_dtype == VC_F32) rc = launch(frames_transform_kernel<float>, static_cast<float*>(out));
  else if (out_dtype == VC_BF16) rc = launch(frames_transform_kernel<bf16>, static_cast<bf16*>(out));
  else rc = launch(frames_transform_kernel<uint8_t>, static_cast<uint8_t*>(out));
  if (rc != VC_OK) return rc;
  VC_LAUNCH_CHECK("vc_frames_transform");
  return VC_OK;
}

int vc_frames_jitter(void* planes_u8, int n, int h, int w, const void* order, const float* factors,
                     const void* hue_shift, const float* mean, const float* stdv, void* out, int out_dtype,
                     hipStream_t stream) {
  VC_CHECK_SHAPE(n >= 0 && h >= 1 && w >= 1 && (int64_t)h * w * 3 < (1ll << 31), "vc_frames_jitter: bad extents");
  VC_CHECK_ARG(out_dtype == VC_F32 || out_dtype == VC_BF16, "vc_frames_jitter: dtype %d", out_dtype);
  if (n == 0) return VC_OK;
  VC_CHECK_ARG(planes_u8 && order && factors && hue_shift && mean && stdv && out, "vc_frames_jitter: null pointer");
  uint8_t* p = static_cast<uint8_t*>(planes_u8);
  const uint8_t* ord = static_cast<const uint8_t*>(order);
  const uint8_t* hs = static_cast<const uint8_t*>(hue_shift);
  if (out_dtype == VC_F32)
    hipLaunchKernelGGL(frames_jitter_kernel<float>, dim3(n), dim3(JT_THREADS), 0, stream, p, h * w, ord, factors, hs,
                       mean, stdv, static_cast<float*>(out));
  else
    hipLaunchKernelGGL(frames_jitter_kernel<bf16>, dim3(n), dim3(JT_THREADS), 0, stream, p, h * w, ord, factors, hs,
                       mean, stdv, static_cast<bf16*>(out));
  VC_LAUNCH_CHECK("vc_frames_jitter");
  return VC_OK;
}

}  // extern "C"
