// Device-side helpers shared by the implicit-GEMM translation units (vc_igemm.hip, vc_igemm8.hip): pixel-row decode,
// gather addressing, LDS-DMA primitives and the epilogues of the NT (forward / data-gradient) kernels.
#pragma once
#include <type_traits>

#include "vc_common.h"

namespace {

constexpr int BM = 128;  // pixels per tile
constexpr int BN = 128;  // output channels per tile
constexpr int BK = 64;   // reduction elements per LDS stage
constexpr int NTHR = 256;

// ----------------------------------------------------------------------------
// Pixel-row decode and gather addressing shared by all kernels.
// ----------------------------------------------------------------------------
struct RowPos {
  int n, d0, h0, w0;  // MODE 0: od*sd-pd, ...   MODE 1: od+pd, ...
};

template <int MODE>
__device__ __forceinline__ RowPos decode_row(const ConvGeom& g, int m) {
  RowPos r;
  if (m >= g.M) {
    r.n = 0;
    r.d0 = r.h0 = r.w0 = -(1 << 28);
    return r;
  }
  int t, ow, oh, od;
  fd_divmod(m, g.fWo, t, ow);
  fd_divmod(t, g.fHo, t, oh);
  fd_divmod(t, g.fDo, t, od);
  r.n = t;
  if (MODE == 0) {
    r.d0 = od * g.sd - g.pd;
    r.h0 = oh * g.sh - g.ph;
    r.w0 = ow * g.sw - g.pw;
  } else {  // MODE 1 and 2
    r.d0 = od + g.pd;
    r.h0 = oh + g.ph;
    r.w0 = ow + g.pw;
  }
  return r;
}

// output pixel (row of the output matrix) of GEMM row m in a phase view
__device__ __forceinline__ int phase_out_pixel(const ConvGeom& g, int m) {
  int t, wc, hc, od;
  fd_divmod(m, g.fWo, t, wc);
  fd_divmod(t, g.fHo, t, hc);
  fd_divmod(t, g.fDo, t, od);
  return ((t * g.Do + od) * g.oHo + g.o_h0 + g.o_hs * hc) * g.oWo + g.o_w0 + g.o_ws * wc;
}

// Offset (in elements) of the source pixel for row r and tap (rd, rh, rw); -1 if the
// tap falls outside the source (zero padding / no contributing output).
template <int MODE>
__device__ __forceinline__ int gather_pixel(const ConvGeom& g, const RowPos& r, int rd, int rh, int rw) {
  int id, ih, iw;
  if (MODE == 2) {
    id = r.d0 - rd;
    ih = r.h0 - rh;
    iw = r.w0 - rw;
    if ((unsigned)id >= (unsigned)g.Di || (unsigned)ih >= (unsigned)g.Hi || (unsigned)iw >= (unsigned)g.Wi) return -1;
  } else if (MODE == 0) {
    id = r.d0 + rd;
    ih = r.h0 + rh;
    iw = r.w0 + rw;
    if ((unsigned)id >= (unsigned)g.Di || (unsigned)ih >= (unsigned)g.Hi || (unsigned)iw >= (unsigned)g.Wi) return -1;
  } else {
    const int td = r.d0 - rd, th = r.h0 - rh, tw = r.w0 - rw;
    if ((td | th | tw) < 0) return -1;
    if ((td & (g.sd - 1)) | (th & (g.sh - 1)) | (tw & (g.sw - 1))) return -1;
    id = td >> (__builtin_ctz(g.sd));
    ih = th >> (__builtin_ctz(g.sh));
    iw = tw >> (__builtin_ctz(g.sw));
    if (id >= g.Di || ih >= g.Hi || iw >= g.Wi) return -1;
  }
  return ((r.n * g.Di + id) * g.Hi + ih) * g.Wi + iw;
}

__device__ __forceinline__ void decode_chunk(const ConvGeom& g, int kc, int& rd, int& rh, int& rw, int& c) {
  int tap, c8, t2;
  fd_divmod(kc, g.fC8, tap, c8);
  c = c8 * 8;
  fd_divmod(tap, g.fkw, t2, rw);
  fd_divmod(t2, g.fkh, rd, rh);
}

// column (element offset inside a packed weight row) of k-chunk kc = (rd, rh, rw, c)
template <int MODE>
__device__ __forceinline__ int weight_col(const ConvGeom& g, int kc, int rd, int rh, int rw, int c) {
  if (MODE != 2) return kc * 8;
  return ((rd * g.wt_kh + g.wt_h0 + g.wt_hs * rh) * g.wt_kw + g.wt_w0 + g.wt_ws * rw) * g.Ci + c;
}

// ----------------------------------------------------------------------------
// NT kernel (forward / data gradient / linear).
// ----------------------------------------------------------------------------
__device__ __forceinline__ int nt_lds_off(int row, int chunk) {  // in elements
  return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

__device__ __forceinline__ void xcd_remap(int& logical) {
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
  logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// 64 bytes of zeros: the source of LDS-DMA lanes whose tap is padding / out of range
__device__ __attribute__((aligned(64))) const uint32_t vc_zero_src[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One wave-instruction of LDS-DMA: every lane moves 16 bytes from its own global address to
// lds_base + lane*16 (the destination is lane-linear by construction of the hardware).
__device__ __forceinline__ void glds16(const void* g, bf16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// Bit t of the result = filter tap t (kw fastest, then kh, then kd) of the window whose origin is r lies inside
// the gather source.  The three axes are tested separately (kd + kh + kw compares) and combined.
template <int MODE>
__device__ __forceinline__ unsigned tap_mask(const ConvGeom& g, const RowPos& r) {
  if (g.kd * g.kh * g.kw == 1)  // 1x1 convolutions and linear layers: one test, no loops
    return (unsigned)(((unsigned)r.d0 < (unsigned)g.Di) & ((unsigned)r.h0 < (unsigned)g.Hi) & ((unsigned)r.w0 < (unsigned)g.Wi));
  unsigned dm = 0, hm = 0, wmk = 0;
  for (int t = 0; t < g.kd; ++t) dm |= (unsigned)((unsigned)(MODE == 0 ? r.d0 + t : r.d0 - t) < (unsigned)g.Di) << t;
  for (int t = 0; t < g.kh; ++t) hm |= (unsigned)((unsigned)(MODE == 0 ? r.h0 + t : r.h0 - t) < (unsigned)g.Hi) << t;
  for (int t = 0; t < g.kw; ++t) wmk |= (unsigned)((unsigned)(MODE == 0 ? r.w0 + t : r.w0 - t) < (unsigned)g.Wi) << t;
  unsigned mk = 0;
  int sh = 0;
  for (int a = 0; a < g.kd; ++a)
    for (int b = 0; b < g.kh; ++b, sh += g.kw)
      if ((dm >> a) & (hm >> b) & 1u) mk |= wmk << sh;
  return mk;
}

// Epilogue shared by the NT kernels: a lane holds 4 consecutive channels of one pixel per 16x16 tile.
template <int MODE, int WI = 4>
__device__ __forceinline__ void nt_epilogue(const ConvGeom& g, f32x4 (&acc)[WI][4], int m0, int n0, int wm, int wn,
                                            int lane, const float* __restrict__ bias, void* __restrict__ out,
                                            int relu, int out_f32, int beta) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + (lane & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int co = n0 + wn * (WI * 16) + i * 16 + (lane >> 4) * 4;
      if (co >= g.Co) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      const size_t o = (size_t)(MODE == 2 ? phase_out_pixel(g, m) : m) * g.ldc + co;
      const bool full = (co + 3 < g.Co);
      if (bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < g.Co) v[r] += bias[co + r];
      }
      if (out_f32) {
        float* op = reinterpret_cast<float*>(out) + o;
        if (beta) {
          if (full && ((g.ldc & 3) == 0)) {
            const f32x4 old = *reinterpret_cast<const f32x4*>(op);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += old[r];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (co + r < g.Co) v[r] += op[r];
          }
        }
        if (relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (full && ((g.ldc & 3) == 0)) {
          *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co + r < g.Co) op[r] = v[r];
        }
      } else {
        bf16* op = reinterpret_cast<bf16*>(out) + o;
        if (beta) {
          if (full && ((g.ldc & 3) == 0)) {
            const bf16x4 old = *reinterpret_cast<const bf16x4*>(op);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (co + r < g.Co) v[r] += (float)op[r];
          }
        }
        if (relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (full && ((g.ldc & 3) == 0)) {
          *reinterpret_cast<bf16x4*>(op) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co + r < g.Co) op[r] = (bf16)v[r];
        }
      }
    }
  }
}

// Epilogue through LDS for compute-type (bf16) outputs: the wave parks its 64x64 sub-tile in an XOR-swizzled
// LDS image and writes it out row-major, 16 bytes per lane, so that every store instruction covers whole
// 128-byte row segments (the accumulator layout alone gives 32-byte pieces of 16 different rows).
// `stage_buf` is this wave's private 8 KB region; the caller has made sure nobody still reads it.
//
// EPI selects what else the epilogue does with a side tensor `tail_y` of the output's own [row][Co] layout
// (requires ldc == Co, Co % 8 == 0):
// EPI_BN_TAIL: the output is the gradient of a residual block's output out = relu(bn(y) + shortcut), complete once
//   this tile has been added (beta) or written.  The epilogue also applies that block's recorded ReLU decisions
//   (tail_bits [row][Co/8], one byte per 8 channels), stores the masked gradient dz, and writes the slab sums the
//   BatchNorm backward needs, bn_partial[slab][0][c] = sum dz, [slab][1][c] = sum dz * y (y = tail_y, the
//   BatchNorm's input).
// EPI_RESIDUAL: out = result + tail_y (a transformer block's x + f(x), written out of place).
// EPI_GELU_BWD: out = result * gelu'(tail_y) (the gradient through a GELU whose input tail_y was kept).
// EPI_GELU_DUAL: out = result, and gelu(result as stored) goes to a second tensor of the same layout whose address
// arrives in tail_y (the only mode that writes through that pointer).
constexpr int EPI_NONE = 0, EPI_BN_TAIL = 1, EPI_RESIDUAL = 2, EPI_GELU_BWD = 3, EPI_GELU_DUAL = 4, EPI_BN_PRE = 5;
// EPI_BN_PRE: the tile is the gradient of relu(bn(y)); `bias` carries the BatchNorm's [4][G][Co] table (mean, invstd,
// scale, shift per group of `relu` = rows-per-group rows), tail_y = y.  The epilogue zeroes the gradient where
// y * scale + shift <= 0 and writes the slab sums of dz and dz * y, as EPI_BN_TAIL does from recorded decisions.
__device__ __forceinline__ float gelu_dev(float x) { return vc_gelu(x); }
__device__ __forceinline__ float gelu_grad_dev(float x) { return vc_gelu_grad(x); }

// 16-byte store of an output-tile row: non-temporal, the tile is not read again by this kernel and the consumers
// stream it from memory anyway (measured per step, stores only: ResNet-50 CRNN 46.30 -> 46.13 ms, Swin-RNN with 32
// clips 53.64 -> 52.75 ms, with 8 clips 17.93 -> 17.85 ms; non-temporal loads of the epilogues' side tensors on top:
// +0.1 ms on the ResNet step)
__device__ __forceinline__ void epi_store8(bf16* p, const bf16x8& v) {
#ifdef VC_ABL_NOSTORE  // ablation build: the tile is not written (time only)
  if (reinterpret_cast<uintptr_t>(p) == 1) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p));
#else
  __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p));
#endif
}

template <int MODE, int WI, int EPI = EPI_NONE, bool HAS_BIAS = true>
__device__ __forceinline__ void nt_epilogue_lds(const ConvGeom& g, f32x4 (&acc)[WI][4], int m0, int n0, int wm,
                                                int wn, int lane, const float* __restrict__ bias,
                                                bf16* __restrict__ out, int relu, int beta, bf16* stage_buf,
                                                float* __restrict__ bn_partial,
                                                const bf16* __restrict__ tail_y = nullptr,
                                                const uint8_t* __restrict__ tail_bits = nullptr) {
  constexpr int WC = WI * 16;   // channels per wave (64 or 32)
  constexpr int NCH = WC / 8;   // 16-byte chunks per image row (8 or 4)
  const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int col = i * 16 + q * 4;  // channel inside the wave's WC
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (HAS_BIAS && bias && EPI != EPI_BN_PRE) {
      const int co = n0 + wn * WC + col;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (co + r < g.Co) b[r] = bias[co + r];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = j * 16 + l15;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] + b[r];
        if (relu && !beta && EPI != EPI_BN_PRE && EPI != EPI_RESIDUAL) v[r] = fmaxf(v[r], 0.f);
      }
      const int chunk = (col >> 3) ^ (row & (NCH - 1));
      *reinterpret_cast<bf16x4*>(stage_buf + row * WC + chunk * 8 + (col & 7)) =
          bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    }
  }
  // the image is private to the wave: its own LDS writes complete before its reads (lgkmcnt order)
  constexpr int RPI = 64 / NCH;  // image rows per store instruction (8 or 16)
  const int rr = lane / NCH, ch = lane % NCH;
  const int co = n0 + wn * WC + ch * 8;
  // per-channel sum / sum of squares of the values as stored (BatchNorm statistics).  They cost the write-heavy
  // forward layers up to a third of their time (64 -> 256 channels at 56 x 56: 0.246 -> 0.327 ms, about 0.85 ms per
  // ResNet-50 step over all layers) and are still cheaper than a pass over y.  Two cheaper-looking forms measured
  // SLOWER: channel pairs on packed fp32 instructions (that layer 0.379 ms, step +1.1 ms), and a fold in which the
  // lanes exchange halves (14 exchanges instead of 48; 48 bytes of scratch at the 256-register limit, step +3.6 ms).
  // The rest runs in two copies: one for tiles that lie wholly inside the matrix with 16-byte-aligned rows (no per-row
  // or per-channel predicates: the tests and the branches around every store are a third of the epilogue's
  // instructions), one with every test for the edge tiles.
  auto finish = [&](auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    float ps[8], pq[8];
  #pragma unroll
    for (int e = 0; e < 8; ++e) ps[e] = pq[e] = 0.f;
    constexpr int NIT = 64 / RPI;
    const bool vec = FULL || (co + 7 < g.Co && (g.ldc & 7) == 0);
    // beta: fetch everything the tile accumulates into up front (one memory latency, not one per row group)
    bf16x8 oldv[NIT];
    bf16* orow[NIT];
    constexpr bool TAIL = EPI == EPI_BN_TAIL;
    constexpr bool PRE = EPI == EPI_BN_PRE;
    constexpr bool SIDE_IN = EPI != EPI_NONE && EPI != EPI_GELU_DUAL;  // tail_y is read
    float psc[PRE ? 8 : 1], psh[PRE ? 8 : 1];
    if (PRE) {  // host guarantees vec, beta == 0 and 64-row slabs inside one group
      const int rpg = relu, G = g.M / rpg;
      const int ms = m0 + wm * 64;
      const int gi = FULL || ms < g.M ? ms / rpg : 0;
  #pragma unroll
      for (int e = 0; e < 8; ++e) {
        psc[e] = FULL || co + e < g.Co ? bias[(size_t)(2 * G + gi) * g.Co + co + e] : 0.f;
        psh[e] = FULL || co + e < g.Co ? bias[(size_t)(3 * G + gi) * g.Co + co + e] : 0.f;
      }
    }
    bf16x8 tyv[SIDE_IN ? NIT : 1];
    unsigned tbits[TAIL ? NIT : 1];
  #pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m0 + wm * 64 + it * RPI + rr;
      const bool live = FULL || (m < g.M && co < g.Co);
      orow[it] = live ? out + (size_t)(MODE == 2 ? phase_out_pixel(g, m) : m) * g.ldc + co : nullptr;
      oldv[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (beta && live && vec) oldv[it] = *reinterpret_cast<const bf16x8*>(orow[it]);
      if (SIDE_IN) {
        tyv[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (live) tyv[it] = *reinterpret_cast<const bf16x8*>(tail_y + (size_t)m * g.Co + co);
      }
      if (TAIL) {
        tbits[it] = 0;
        if (live) tbits[it] = tail_bits[((size_t)m * g.Co + co) >> 3];
      }
    }
  #pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * RPI + rr;
      const int m = m0 + wm * 64 + row;
      bf16x8 v = *reinterpret_cast<const bf16x8*>(stage_buf + row * WC + ((ch ^ (row & (NCH - 1))) << 3));
      if (!TAIL && !PRE && bn_partial && (FULL || m < g.M)) {
  #pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = (float)v[e];
          ps[e] += t;
          pq[e] = fmaf(t, t, pq[e]);
        }
      }
      bf16* op = orow[it];
      if (TAIL) {  // host guarantees vec; rows beyond M carry zero bits and contribute nothing
  #pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = (float)v[e];
          if (beta) t += (float)oldv[it][e];
          v[e] = ((tbits[it] >> e) & 1) ? (bf16)t : (bf16)0.f;
          const float dz = (float)v[e];
          ps[e] += dz;
          pq[e] = fmaf(dz, (float)tyv[it][e], pq[e]);
        }
        if (FULL || op) epi_store8(op, v);
      } else if (PRE) {
  #pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yy = (float)tyv[it][e];
          v[e] = fmaf(yy, psc[e], psh[e]) > 0.f ? v[e] : (bf16)0.f;
          const float dz = (float)v[e];  // rows beyond M: y was read as zero and dz counts for nothing (not stored)
          if (FULL || m < g.M) {
            ps[e] += dz;
            pq[e] = fmaf(dz, yy, pq[e]);
          }
        }
        if (FULL || op) epi_store8(op, v);
      } else if (EPI == EPI_GELU_DUAL) {  // host guarantees vec and beta == 0
        if (FULL || op) {
          bf16x8 gv;
  #pragma unroll
          for (int e = 0; e < 8; ++e) gv[e] = (bf16)gelu_dev((float)v[e]);
          epi_store8(op, v);
          epi_store8(const_cast<bf16*>(tail_y) + (size_t)m * g.Co + co, gv);
        }
      } else if (EPI == EPI_RESIDUAL || EPI == EPI_GELU_BWD) {  // host guarantees vec
        if (FULL || op) {
          // EPI_RESIDUAL with a row scale (stochastic depth): tail_bits carries fp32 scales, one per `relu` rows;
          // without one, relu < 0 asks for a ReLU AFTER the residual has been added
          float rs = 1.f;
          if (EPI == EPI_RESIDUAL && tail_bits) rs = reinterpret_cast<const float*>(tail_bits)[m / relu];
          const bool relu_sum = EPI == EPI_RESIDUAL && relu < 0;
  #pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = (float)v[e];
            if (beta) t += (float)oldv[it][e];
            if (EPI == EPI_RESIDUAL) {
              t = fmaf(t, rs, (float)tyv[it][e]);
              if (relu_sum) t = fmaxf(t, 0.f);
            } else {
              t *= gelu_grad_dev((float)tyv[it][e]);
            }
            v[e] = (bf16)t;
          }
          epi_store8(op, v);
        }
      } else if (FULL || op) {
        if (vec) {
          if (beta) {
  #pragma unroll
            for (int e = 0; e < 8; ++e) {
              float t = (float)v[e] + (float)oldv[it][e];
              if (relu) t = fmaxf(t, 0.f);
              v[e] = (bf16)t;
            }
          }
          epi_store8(op, v);
        } else {
  #pragma unroll
          for (int e = 0; e < 8; ++e)
            if (co + e < g.Co) {
              float t = (float)v[e];
              if (beta) {
                t += (float)op[e];
                if (relu) t = fmaxf(t, 0.f);
              }
              op[e] = (bf16)t;
            }
        }
      }
    }
    if (bn_partial && WI == 4 && !TAIL) {
      // The lanes that hold the same channels differ in rr (8 row groups).  Their partial sums go through the wave's
      // image (free by now, 8 row groups x 64 channels x {sum, sum of squares} = 4 KiB): one LDS round trip and 14
      // additions instead of three dependent cross-lane exchanges of 16 values, and lane l ends up with channel l, so
      // the slab's row bn_partial[slab][0 / 1][64 channels] is written by two whole-wave stores.  (Not for the
      // batch-norm backward tail: at its 256 registers this form costs 88 bytes of scratch per lane.)
      float* fs = reinterpret_cast<float*>(stage_buf);
      f32x4* w = reinterpret_cast<f32x4*>(fs + rr * 64 + ch * 8);
      w[0] = f32x4{ps[0], ps[1], ps[2], ps[3]};
      w[1] = f32x4{ps[4], ps[5], ps[6], ps[7]};
      w[128] = f32x4{pq[0], pq[1], pq[2], pq[3]};  // + 512 floats
      w[129] = f32x4{pq[4], pq[5], pq[6], pq[7]};
      float s = 0.f, q = 0.f;
  #pragma unroll
      for (int r = 0; r < 8; ++r) {
        s += fs[r * 64 + lane];
        q += fs[512 + r * 64 + lane];
      }
      const int cl = n0 + wn * WC + lane;
      if (FULL || (cl < g.Co && m0 + wm * 64 < g.M)) {
        float* d = bn_partial + (size_t)((m0 + wm * 64) >> 6) * 2 * g.Co + cl;
        d[0] = s;
        d[g.Co] = q;
      }
    } else if (bn_partial) {
      // fold the lanes that hold the same channels (they differ in rr) and let rr == 0 write the slab's row:
      // bn_partial[slab][0][co] = sum, [slab][1][co] = sum of squares over the slab's 64 pixel rows
  #pragma unroll
      for (int e = 0; e < 8; ++e) {
  #pragma unroll
        for (int o = NCH; o < 64; o <<= 1) {
          ps[e] += __shfl_xor(ps[e], o, 64);
          pq[e] += __shfl_xor(pq[e], o, 64);
        }
      }
      if (rr == 0 && (FULL || (co < g.Co && m0 + wm * 64 < g.M))) {
        const int slab = (m0 + wm * 64) >> 6;
        float* d = bn_partial + (size_t)slab * 2 * g.Co + co;
  #pragma unroll
        for (int e = 0; e < 8; ++e)
          if (FULL || co + e < g.Co) {
            d[e] = ps[e];
            d[g.Co + e] = pq[e];
          }
      }
    }
  };
  const bool full_tile = MODE != 2 && m0 + wm * 64 + 64 <= g.M && n0 + wn * WC + WC <= g.Co && (g.ldc & 7) == 0 &&
                         (g.Co & 7) == 0;
  // (64-channel wave tiles only: with the second copy the 64-wide kernel needs 176 registers instead of 168 and loses
  // its third workgroup per CU; the batch-norm backward tail keeps one copy as well, two do not fit its 256 registers)
  if (WI == 4 && EPI != EPI_BN_TAIL && full_tile) finish(std::true_type{});
  else finish(std::false_type{});
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global access
// (vmcnt(0)); where only LDS reads have to be over, this one lets the LDS-DMA stage in flight keep streaming.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace

// 256 x 256-tile, 8-wave form of the NT kernel (vc_igemm8.hip).  `mode` 0 forward, 1 data gradient, 2 one output phase of
// a strided data gradient; `epi` one of the EPI_* constants; bf16 output only.
bool vc_nt8_eligible(const ConvGeom& g, int mode, int out_f32, int epi);
void vc_nt8_launch(const ConvGeom& g, int mode, int epi, hipStream_t stream, const void* src, const void* wgt,
                   const float* bias, void* out, int relu, int beta, float* bn_partial, const void* tail_y,
                   const uint8_t* tail_bits);

// 256 x 256-tile, 8-wave form of the weight-gradient kernel (vc_wgrad8.hip): bf16.
// vc_wgrad8_plan chooses the slices of the pixel rows (m_per_split a multiple of 64, or 0: slice s works k-tiles s,
// s + splits, ... of 64 rows); the slices are added with float
// atomics into dw, or into copies `split_stride` floats apart when that is not zero.  dbias (or null): the column sums
// of dy are added to it with float atomics (not with split_stride != 0).
bool vc_wgrad8_eligible(const ConvGeom& g, int ldy, bool with_bias);
void vc_wgrad8_plan(const ConvGeom& g, size_t det_capacity, bool det, int* splits, int* m_per_split);
void vc_wgrad8_launch(const ConvGeom& g, const void* x, const void* dy, int ldy, float* dw, int splits, int m_per_split,
                      size_t split_stride, float* dbias, hipStream_t stream);
