// Weight gradient of a convolution / linear layer on 256 x 256 output tiles (output channels x gather columns), gfx950.
//
//   dw[co][col] += sum over pixel rows m of dy[m][co] * xg[m][col],   col = (tap, input channel), xg = the gathered x.
//
// One workgroup of 8 waves works one output tile over one slice of the pixel rows (the slices of a tile are added with
// float atomics, or into per-slice copies that are added in order afterwards in deterministic mode).  On the full tile
// the waves form a 2 x 4 grid, each owning 128 output channels x 64 columns = 8 x 4 accumulator tiles of
// v_mfma_f32_16x16x32_bf16; narrower tiles (256 x 128, 128 x 256, 256 x 64, 64 x 256) serve layers with fewer output
// channels or columns.  Both operands are staged pixel-major ([64 rows][tile side], rows of 512 / 256 / 128 bytes; two
// stages of 64 KiB for the full tile, three for the narrower ones) by LDS-DMA requests of 1 KiB = 2 / 4 / 8 whole rows
// -- contiguous runs of dy and of x -- and read with the transposing ds_read_b64_tr_b16, since the reduction index
// runs along the LDS rows.
//
// Schedule: that of the 256-wide NT kernel (vc_igemm8.hip).  A k-tile of 64 pixel rows is two halves of 32; while a
// wave multiplies one half out of one fragment set (8 + 4 operand registers quads) it reads the other half, or the first
// half of the next k-tile, into a second set and issues its 8 requests of the k-tile after next between the products;
// one barrier per k-tile.  Requests are opaque assembly (see vc_igemm8.hip for why).
//
// Gather: a lane's column (tap, channel chunk) is fixed for the whole kernel; the pixel rows of a k-tile are decoded
// once, three k-tiles ahead, by 64 threads into a table in LDS (window origin offset and coordinates); a lane tests its
// tap against the row's origin and presents an offset beyond the descriptor's range where the tap is padding or the
// row lies outside the slice: the hardware writes zeros.  dy rows beyond the slice fall outside the dy descriptor,
// which ends at the slice's last row.
#include <stdlib.h>

#include <stdio.h>

#include "vc_igemm_dev.h"

namespace {

constexpr int NW8 = 512;   // threads per workgroup
constexpr int RB = 64;     // pixel rows per k-tile

typedef int i32x4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wdma16(const i32x4w& rs, unsigned lds_byte, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff)
               : "memory");
}
__device__ __forceinline__ unsigned wlds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

struct RowOrigin {
  int off;         // byte offset of the row's window origin (tap 0,0,0) in x; may be negative
  int d0, h0, w0;  // window origin coordinates; h0 = -2^28 marks a row beyond the slice
};

// Block swizzle of a pixel-major tile whose rows are W elements (W / 16 blocks of 16 columns = 32 bytes): XORed into the
// block index so that the 8 rows one transposing read serves (8 gq + q, and the same + 4) fall on 8 different groups of
// 8 banks.  Rows of 512 or 256 bytes start on the same bank; a 128-byte row is half a bank sweep.
template <int W>
__device__ __forceinline__ int wswz(int row) {
  if (W >= 128) return (row & 3) | ((row >> 1) & 4);
  return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
}

// Tile = (WN x NI x 16) output channels x (WM x NJ x 16) columns, WN x WM = 8 waves, each NI x NJ accumulator tiles.
template <int NI, int NJ, int WN, int WM, bool BIAS>
__global__ __launch_bounds__(NW8, 2) void igemm_tn_wgrad8_kernel(ConvGeom g, const bf16* __restrict__ x,
                                                                 const bf16* __restrict__ dy, int ldy,
                                                                 float* __restrict__ dw, int m_per_split, int ntiles,
                                                                 int tiles_col, size_t split_stride,
                                                                 float* __restrict__ dbias) {
  static_assert(WN * WM == 8, "eight waves");
  constexpr int TCO = WN * NI * 16, TCOL = WM * NJ * 16;
  static_assert((TCO == 64 || TCO == 128 || TCO == 256) && (TCOL == 64 || TCOL == 128 || TCOL == 256), "tile sides");
  constexpr int YSZ = RB * TCO, XSZ = RB * TCOL;  // elements of the operand tiles
  constexpr int STG = YSZ + XSZ;                  // elements per stage: dy tile, then x tile
  // a 1 KiB request covers RPR rows; a wave issues NP requests per operand and k-tile
  constexpr int RPRY = 512 / TCO, RPRX = 512 / TCOL, NPY = RB / (RPRY * 8), NPX = RB / (RPRX * 8), NP = NPY + NPX;
  constexpr int NM = NI * NJ;                                  // products per half k-tile and wave
  constexpr int NF = NI + NJ;                                  // fragments per half
  constexpr int FSTEP = NM >= 2 * NF ? 2 : 1;                  // a fragment after every FSTEP-th product
  constexpr int PSTEP = NM >= 4 * NP ? 4 : NM >= 2 * NP ? 2 : 1;  // a request after every PSTEP-th product
  static_assert(NF * FSTEP <= NM && NP * PSTEP <= NM, "the reads and requests of a half fit between its products");
  // stages: as many as fit beside the row table (2 for the full tile, 3 for the narrower ones, whose k-tiles are short
  // against the memory latency: with two stages every k-tile would wait out a whole round trip)
  constexpr int NS = (160 * 1024 - 2048) / (STG * 2) >= 3 ? 3 : 2;
  __shared__ __attribute__((aligned(1024))) bf16 smem[NS * STG];
  __shared__ __attribute__((aligned(16))) RowOrigin rinfo[2][RB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / WM, wm = wave % WM;  // wn: block of output channels, wm: block of columns
  int logical;
  xcd_remap(logical);
  const int split = logical / ntiles, tile = logical - split * ntiles;
  if (split_stride) dw += (size_t)split * split_stride;  // deterministic mode: one copy of dw per slice
  const int tile_co = tile / tiles_col, tile_col = tile - tile_co * tiles_col;
  const int co0 = tile_co * TCO, col0 = tile_col * TCOL;
  // m_per_split > 0: slice `split` is the contiguous row range [split * m_per_split, ...); m_per_split == 0: the slices
  // are interleaved, slice `split` works k-tiles split, split + S, split + 2 S, ... of the whole matrix
  const bool il = m_per_split == 0;
  const int nsplit = gridDim.x / ntiles;
  const int kstep = il ? nsplit : 1;
  const int m_begin = il ? split * RB : split * m_per_split;
  const int m_end = il ? g.M : min(g.M, m_begin + m_per_split);
  const int total_kt = (g.M + RB - 1) / RB;
  const int nkt = il ? (split < total_kt ? (total_kt - split + nsplit - 1) / nsplit : 0)
                     : (m_begin < m_end ? (m_end - m_begin + RB - 1) / RB : 0);

  // staging roles: request q of a wave covers rows (8 q + wave) RPR + (0 .. RPR-1); lane -> row lr of those, 16-byte
  // piece pc of the row.  The block swizzle of a lane's row does not depend on q (rows 8 RPR >= 16 apart), so the
  // chunk of the source row it fetches is fixed.
  constexpr unsigned SENT = 0xFFFFFFF0u;
  const int ylr = lane / (TCO / 8), ypc = lane % (TCO / 8), ysrow = wave * RPRY + ylr;
  const int ychunk = ((((ypc >> 1) ^ wswz<TCO>(ysrow)) << 1) | (ypc & 1));
  const int xlr = lane / (TCOL / 8), xpc = lane % (TCOL / 8), xsrow = wave * RPRX + xlr;
  const int xchunk = ((((xpc >> 1) ^ wswz<TCOL>(xsrow)) << 1) | (xpc & 1));
  const bool covalid = co0 + ychunk * 8 < g.Co;
  const int col = col0 + xchunk * 8;
  const bool colvalid = col < g.K;
  int rd, rh, rw, c;
  decode_chunk(g, col >> 3, rd, rh, rw, c);
  const int tapoff = (((rd * g.Hi + rh) * g.Wi + rw) * g.ldi + c) * 2;  // bytes
  const int taps_packed = rd | (rh << 8) | (rw << 16);
  // dy: rows of this slice only (the descriptor ends at the slice's last row: later rows read as zeros)
  const i32x4w rs_y = {(int)(uintptr_t)dy, (int)((uintptr_t)dy >> 32), (int)((unsigned)m_end * (unsigned)ldy * 2u),
                       0x00020000};
  const i32x4w rs_x = {(int)(uintptr_t)x, (int)((uintptr_t)x >> 32),
                       (int)((unsigned)(g.N * g.Di * g.Hi * g.Wi) * (unsigned)g.ldi * 2u), 0x00020000};
  const unsigned yvo = covalid ? (unsigned)(((m_begin + ysrow) * ldy + co0 + ychunk * 8) * 2) : SENT;

  auto decode_rows = [&](int kt, int t) {  // row t of k-tile kt into table kt & 1 (read by that k-tile's requests)
    const int m = m_begin + kt * kstep * RB + t;
    RowOrigin ri;
    ri.off = 0;
    ri.d0 = ri.w0 = 0;
    ri.h0 = -(1 << 28);
    if (m < m_end) {
      const RowPos rp = decode_row<0>(g, m);
      ri.d0 = rp.d0;
      ri.h0 = rp.h0;
      ri.w0 = rp.w0;
      ri.off = ((((rp.n * g.Di + rp.d0) * g.Hi + rp.h0) * g.Wi + rp.w0) * g.ldi) * 2;
    }
    rinfo[kt & 1][t] = ri;
  };
  // request p (0 .. NP-1) of k-tile kt into stage kt % NS: first the wave's dy requests, then its x requests
  auto stage_piece = [&](int kt, int p) __attribute__((always_inline)) {
    bf16* sY = smem + (kt % NS) * STG;
    if (p < NPY) {
      // (the whole offset goes into the lane register: only that one is certain to be range-checked)
      wdma16(rs_y, wlds_addr(sY + (p * 8 + wave) * RPRY * TCO),
             yvo == SENT ? SENT : yvo + (unsigned)((kt * kstep * RB + p * 8 * RPRY) * ldy * 2), 0);
    } else {
      const int q = p - NPY;
      const int4 ri = *reinterpret_cast<const int4*>(&rinfo[kt & 1][q * 8 * RPRX + xsrow]);
      const int trd = taps_packed & 255, trh = (taps_packed >> 8) & 255, trw = taps_packed >> 16;
      const bool ok = colvalid && (unsigned)(ri.y + trd) < (unsigned)g.Di && (unsigned)(ri.z + trh) < (unsigned)g.Hi &&
                      (unsigned)(ri.w + trw) < (unsigned)g.Wi;
      wdma16(rs_x, wlds_addr(sY + YSZ + (q * 8 + wave) * RPRX * TCOL), ok ? (unsigned)(ri.x + tapoff) : SENT, 0);
    }
  };

  f32x4 acc[NI][NJ];  // [output-channel tile][column tile]
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Bias gradient (BIAS): the column sums of dy are one more product of the dy fragments, with a B operand of ones.  The
  // workgroups of the first column tile form them; of the WM waves that hold the same dy fragments, wave wm takes the
  // output-channel tiles i = wm, wm + WM, ...  (every column of such an accumulator tile carries the same sums).
  constexpr int NB = BIAS ? (NI + WM - 1) / WM : 1;
  f32x4 bacc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = BIAS && dbias != nullptr && tile_col == 0;
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};

  // transposing fragment reads: lane (gq, q, p) reads rows 8 gq + q and + 4 (+ 32 for the second half) of a 16-column
  // block; the block swizzle of those rows is the same.  Address = (base | swizzle bits) ^ (block << 5): the lane part
  // sits in one register per operand, the block enters as an immediate XOR.
  const int gq = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  const int r0 = 8 * gq + fq;
  const unsigned ybase = (unsigned)(r0 * TCO * 2 + fp * 8) | ((unsigned)wswz<TCO>(r0) << 5);
  const unsigned xbase = (unsigned)(YSZ * 2 + r0 * TCOL * 2 + fp * 8) | ((unsigned)wswz<TCOL>(r0) << 5);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  auto frag = [&](unsigned byte, int rowbytes) __attribute__((always_inline)) {
    union {
      struct {
        s16x4 a, b;
      } s;
      bf16x8 v;
    } u;
    const char* p = reinterpret_cast<const char*>(smem) + byte;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * rowbytes));
    return u.v;
  };
  bf16x8 yX[NI], xX[NJ], yY[NI], xY[NJ];
  // fragment r of a half (first the wave's NJ column tiles, then its NI output-channel tiles) of stage st, half ks
  auto ldf = [&](bf16x8(&yf)[NI], bf16x8(&xf)[NJ], int st, int ks, int r) __attribute__((always_inline)) {
    if (r < NJ)
      xf[r] = frag((xbase + (unsigned)((st * STG + ks * 32 * TCOL) * 2)) ^ (unsigned)((wm * NJ + r) << 5), TCOL * 2);
    else
      yf[r - NJ] = frag((ybase + (unsigned)((st * STG + ks * 32 * TCO) * 2)) ^ (unsigned)((wn * NI + (r - NJ)) << 5),
                        TCO * 2);
  };

  // ---- prologue: row table and requests of k-tiles 0 .. NS-1, the table of k-tile NS; k-tile 0 waited for ----
#pragma unroll 1
  for (int u = 0; u < NS; ++u) {
    if (tid < RB) decode_rows(u, tid);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p) stage_piece(u, p);
    __syncthreads();
  }
  if (tid < RB) decode_rows(NS, tid);
  // counted waits: W1 = all but the requests of the last NS - 1 k-tiles, W2 = all but those of the last NS - 2
  constexpr int W1 = (NS - 1) * NP, W2 = (NS - 2) * NP;
  static_assert(W1 <= 16 && W2 <= 8, "wait counts below are spelled out for these");
#define VC_VMWAIT(N_)                                                                              \
  do {                                                                                             \
    if ((N_) == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");         \
    else if ((N_) == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");    \
    else if ((N_) == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");    \
    else if ((N_) == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");    \
    else if ((N_) == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");  \
    else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");                  \
  } while (0)
  static_assert(W1 == 5 || W1 == 6 || W1 == 8 || W1 == 10 || W1 == 12, "prologue wait");
  static_assert(W2 == 0 || W2 == 5 || W2 == 6, "loop wait");
  VC_VMWAIT(W1);
  if (nkt > 0) {
#pragma unroll
    for (int r = 0; r < NF; ++r) ldf(yX, xX, 0, 0, r);
  }

  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt % NS, stn = (kt + 1) % NS;
    // ---- first half: products out of X, second half's fragments into Y ----
#pragma unroll
    for (int n = 0; n < NM; ++n) {
      const int j = n % NJ, i = n / NJ;
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yX[i], xX[j], acc[i][j], 0, 0, 0);
      if (BIAS && j == 0 && do_bias && i % WM == wm)
        bacc[BIAS ? i / WM : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yX[i], ones, bacc[BIAS ? i / WM : 0], 0, 0, 0);
      if (n % FSTEP == 0 && n / FSTEP < NF) ldf(yY, xY, st, 1, n / FSTEP);
      __builtin_amdgcn_sched_barrier(0);
    }
    // every read of this k-tile's stage is complete, and so are this wave's requests of the next k-tile
    VC_VMWAIT(W2);
    // ---- second half: products out of Y, the next k-tile's first fragments into X, requests of the one after ----
    // (k-tiles past the slice: rows marked outside, dy beyond the descriptor -- zeros land in a stage nobody reads)
#pragma unroll
    for (int n = 0; n < NM; ++n) {
      const int j = n % NJ, i = n / NJ;
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yY[i], xY[j], acc[i][j], 0, 0, 0);
      if (BIAS && j == 0 && do_bias && i % WM == wm)
        bacc[BIAS ? i / WM : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yY[i], ones, bacc[BIAS ? i / WM : 0], 0, 0, 0);
      if (n % FSTEP == 0 && n / FSTEP < NF) ldf(yX, xX, stn, 0, n / FSTEP);
      if (n % PSTEP == PSTEP - 1 && n / PSTEP < NP) stage_piece(kt + NS, n / PSTEP);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tid < RB) decode_rows(kt + NS + 1, tid);  // its table was last read by the requests of k-tile kt + NS - 1
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef VC_VMWAIT

#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + (wn * NI + i) * 16 + (lane >> 4) * 4 + r;
      if (co >= g.Co) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int cc = col0 + (wm * NJ + j) * 16 + (lane & 15);
        if (cc < g.K) atomicAdd(dw + (size_t)co * g.K + cc, acc[i][j][r]);
      }
    }
  }
  if (BIAS && do_bias && (lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (i % WM != wm) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wn * NI + i) * 16 + (lane >> 4) * 4 + r;
        if (co < g.Co) atomicAdd(dbias + co, bacc[BIAS ? i / WM : 0][r]);
      }
    }
  }
}

// VC_WGRAD8: 0 never, 1 every geometry the kernel can run, unset: where it measured faster (vc_wgrad8_eligible)
int wgrad8_policy() {
  const char* e = getenv("VC_WGRAD8");
  return e ? (atoi(e) ? 1 : 0) : 2;
}

// tile sides for a layer: the smallest of 64 / 128 / 256 that holds the side, 256 beyond that
void wgrad8_tile(const ConvGeom& g, int* tco, int* tcol) {
  *tco = g.Co <= 64 ? 64 : g.Co <= 128 ? 128 : 256;
  *tcol = g.K <= 64 ? 64 : g.K <= 128 ? 128 : 256;
  // 256 x 128 instead of 256 x 256 where both sides would be full: a slice then adds half the fp32 tile (the atomics of
  // 256 workgroups x 256 KiB are 64 MB per launch, beside 250 MB of operands on the 14 x 14 1 x 1 layers), measured on
  // the ResNet-50 table: 1 x 1 layers of stages 3-4 0.102 -> 0.086 ms, 3 x 3 at 14 x 14 0.154 -> 0.151; the 3 x 3 layers
  // at 7 x 7 (25088 rows) lose (0.154 -> 0.162) and keep the full tile.  VC_WGRAD8_RULE=0 keeps 256 x 256 everywhere.
  {
    const char* r = getenv("VC_WGRAD8_RULE");
    const int v = r ? atoi(r) : 1;
    const bool short_multi_tap = g.kd * g.kh * g.kw > 1 && g.M < 50000;
    if (*tco == 256 && *tcol == 256 && (v == 2 || (v == 1 && !short_multi_tap))) *tcol = 128;
  }
  // VC_WGRAD8_CAP=co,col caps the tile sides (A/B measurements: 256,128 or 128,256 halve the fp32 tile a slice adds)
  if (const char* e = getenv("VC_WGRAD8_CAP")) {
    int a = 0, b = 0;
    if (sscanf(e, "%d,%d", &a, &b) == 2) {
      if ((a == 128 || a == 64) && *tco > a) *tco = a;
      if ((b == 128 || b == 64) && *tcol > b) *tcol = b;
    }
  }
}
bool wgrad8_has(int tco, int tcol) {  // the instantiated shapes
  return (tco == 256 && (tcol == 256 || tcol == 128 || tcol == 64)) || (tcol == 256 && (tco == 128 || tco == 64));
}

}  // namespace

bool vc_wgrad8_eligible(const ConvGeom& g, int ldy, bool with_bias) {
  const int policy = wgrad8_policy();
  if (!policy) return false;
  // Layers with a bias (linear layers: one tap) form the bias gradient inside the kernel as well (not in deterministic
  // mode: there it is a column-sum pass over dy of its own, and the 128-wide kernel keeps them).  Measured on the Swin-T
  // layers at 128 frames against the 128-wide kernel: the MLP layers (one side four times the other) gain 8-20 % at
  // every stage, 96 -> 288 gains 23 %, the other qkv and the square projections lose 5-20 %.
  if (with_bias && policy != 1) {
    const int lo = g.Co < g.K ? g.Co : g.K, hi = g.Co < g.K ? g.K : g.Co;
    if (vc_deterministic() || g.kd * g.kh * g.kw != 1 || g.M < 4096 || !(hi >= 4 * lo || (lo <= 96 && hi >= 256)))
      return false;
  }
  if (g.Ci % 8 || g.Co % 8 || g.ldi % 8 || ldy % 8) return false;
  if (g.kd > 255 || g.kh > 255 || g.kw > 255) return false;
  const int64_t xbytes = (int64_t)g.N * g.Di * g.Hi * g.Wi * g.ldi * 2, ybytes = (int64_t)g.M * ldy * 2;
  if (xbytes >= 0x7FFF0000ll || ybytes >= 0x7FFF0000ll) return false;
  int tco, tcol;
  wgrad8_tile(g, &tco, &tcol);
  if (!wgrad8_has(tco, tcol)) return false;
  if (policy == 1) return true;
  // one workgroup per CU and slices of at least 512 rows: a short matrix (the LSTM and decoder layers, 512 rows) would
  // leave most of the chip idle (CRNN step 5.78 -> 5.69 ms with them on the 128-wide kernel)
  const int ntiles = ceil_div(g.Co, tco) * ceil_div(g.K, tcol), cus = vc_device_cus();
  int slices = cus / ntiles > 0 ? cus / ntiles : 1;
  if (slices > (g.M / 512 > 1 ? g.M / 512 : 1)) slices = g.M / 512 > 1 ? g.M / 512 : 1;
  if (2 * ntiles * slices < cus) return false;
  // measured against the 128-wide kernel on the ResNet-50 layer table (512 frames): faster or equal on every shape
  // except 3 x 3 layers with 64 output channels (0.31 against 0.24 ms: three column tiles re-read dy); the unfolded
  // stem (7 taps of 24 channels, one column tile) gains (0.423 against 0.458 ms)
  return !(tco == 64 && g.kd * g.kh * g.kw > 1 && g.K > 256);
}

// splits: slices of the pixel rows; m_per_split a multiple of 64, or 0 for interleaved slices.  One workgroup per CU and launch where the layer
// allows; never fewer than 512 rows per slice (the fp32 tile a slice adds costs as much as ~100 rows of products).
void vc_wgrad8_plan(const ConvGeom& g, size_t det_capacity, bool det, int* splits, int* m_per_split) {
  int tco, tcol;
  wgrad8_tile(g, &tco, &tcol);
  const int ntiles = ceil_div(g.Co, tco) * ceil_div(g.K, tcol);
  const int cus = vc_device_cus();
  int s = cus / ntiles;
  if (s < 1) s = 1;
  const int cap = g.M / 512 > 1 ? g.M / 512 : 1;
  if (s > cap) s = cap;
  const size_t nw = (size_t)g.Co * g.K;
  if (det && (size_t)s * nw > det_capacity) {
    s = (int)(det_capacity / nw);
    if (s < 1) s = 1;
  }
  const int mps = ceil_div(ceil_div(g.M, s), RB) * RB;
  *splits = ceil_div(g.M, mps);
  // Interleaved slices by default: the workgroups then read one moving window of the two tensors together instead of
  // 256 streams megabytes apart (the layers bound by memory run 7 % faster: 0.239 -> 0.222 ms on 56 x 56 x 64 -> 256).
  // VC_WGRAD8_IL=0 gives every slice one contiguous row range.
  const char* e = getenv("VC_WGRAD8_IL");
  *m_per_split = (e && !atoi(e)) ? mps : 0;
}

void vc_wgrad8_launch(const ConvGeom& g, const void* x, const void* dy, int ldy, float* dw, int splits, int m_per_split,
                      size_t split_stride, float* dbias, hipStream_t stream) {
  int tco, tcol;
  wgrad8_tile(g, &tco, &tcol);
  const int tiles_col = ceil_div(g.K, tcol), ntiles = ceil_div(g.Co, tco) * tiles_col;
#define VC_W8(NI_, NJ_, WN_, WM_)                                                                                      \
  do {                                                                                                                 \
    if (dbias)                                                                                                         \
      hipLaunchKernelGGL((igemm_tn_wgrad8_kernel<NI_, NJ_, WN_, WM_, true>), dim3(ntiles * splits), dim3(NW8), 0,      \
                         stream, g, (const bf16*)x, (const bf16*)dy, ldy, dw, m_per_split, ntiles, tiles_col,          \
                         split_stride, dbias);                                                                         \
    else                                                                                                               \
      hipLaunchKernelGGL((igemm_tn_wgrad8_kernel<NI_, NJ_, WN_, WM_, false>), dim3(ntiles * splits), dim3(NW8), 0,     \
                         stream, g, (const bf16*)x, (const bf16*)dy, ldy, dw, m_per_split, ntiles, tiles_col,          \
                         split_stride, nullptr);                                                                       \
  } while (0)
  if (tco == 256 && tcol == 256) VC_W8(8, 4, 2, 4);
  else if (tco == 256 && tcol == 128) VC_W8(4, 4, 4, 2);
  else if (tco == 128 && tcol == 256) VC_W8(4, 4, 2, 4);
  else if (tco == 256 && tcol == 64) VC_W8(4, 2, 4, 2);
  else VC_W8(2, 4, 2, 4);  // 64 x 256
#undef VC_W8
}
