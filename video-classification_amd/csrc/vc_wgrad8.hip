// Weight gradient of a convolution / linear layer on 256 x 256 output tiles (output channels x gather columns), gfx950.
//
//   dw[co][col] += sum over pixel rows m of dy[m][co] * xg[m][col],   col = (tap, input channel), xg = the gathered x.
//
// One workgroup of 8 waves works one output tile over one slice of the pixel rows (the slices of a tile are added with
// float atomics, or into per-slice copies that are added in order afterwards in deterministic mode).  The waves form a
// 2 x 4 grid, each owning 128 output channels x 64 columns = 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16.
// Both operands are staged pixel-major ([64 rows][256 channels or columns], 512-byte rows, two 64 KiB stages) by
// LDS-DMA requests of 1 KiB = two whole rows -- contiguous 512-byte runs of dy and of x -- and read with the
// transposing ds_read_b64_tr_b16, since the reduction index runs along the LDS rows.
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

#include "vc_igemm_dev.h"

namespace {

constexpr int WT = 256;    // tile width in output channels and in columns
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

// Block swizzle of a pixel-major tile with 512-byte rows (16 blocks of 16 columns): XORed into the block index so that
// the 8 rows one transposing read serves (8 * gq + q, and the same + 4) fall on 8 different groups of 8 banks.
__device__ __forceinline__ int wswz(int row) { return (row & 3) | ((row >> 1) & 4); }

__global__ __launch_bounds__(NW8, 2) void igemm_tn_wgrad8_kernel(ConvGeom g, const bf16* __restrict__ x,
                                                                 const bf16* __restrict__ dy, int ldy,
                                                                 float* __restrict__ dw, int m_per_split, int ntiles,
                                                                 int tiles_col, size_t split_stride) {
  constexpr int TSZ = RB * WT;      // elements of one operand tile (32 KiB)
  constexpr int STG = 2 * TSZ;      // elements per stage: dy tile, then x tile
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * STG];
  __shared__ __attribute__((aligned(16))) RowOrigin rinfo[2][RB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wm = wave & 3;  // wn: half of the output channels, wm: quarter of the columns
  int logical;
  xcd_remap(logical);
  const int split = logical / ntiles, tile = logical - split * ntiles;
  if (split_stride) dw += (size_t)split * split_stride;  // deterministic mode: one copy of dw per slice
  const int tile_co = tile / tiles_col, tile_col = tile - tile_co * tiles_col;
  const int co0 = tile_co * WT, col0 = tile_col * WT;
  const int m_begin = split * m_per_split;
  const int m_end = min(g.M, m_begin + m_per_split);
  const int nkt = m_begin < m_end ? (m_end - m_begin + RB - 1) / RB : 0;

  // staging roles: a request is two rows of 512 bytes; lane -> row lr of the two, 16-byte piece pc of 32.  The block
  // swizzle of the row a lane serves does not depend on the request (rows 16 apart), so its source chunk is fixed.
  constexpr unsigned SENT = 0xFFFFFFF0u;
  const int lr = lane >> 5, pc = lane & 31;
  const int srow = wave * 2 + lr;                                   // row of the 16 a request group covers
  const int chunk = ((((pc >> 1) ^ wswz(srow)) << 1) | (pc & 1));   // 8-element chunk of the 256 this lane fetches
  const bool covalid = co0 + chunk * 8 < g.Co;
  const int col = col0 + chunk * 8;
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
  const unsigned yvo = covalid ? (unsigned)(((m_begin + srow) * ldy + co0 + chunk * 8) * 2) : SENT;

  auto decode_rows = [&](int kt, int t) {  // row t of k-tile kt into table kt & 1
    const int m = m_begin + kt * RB + t;
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
  // request p (0..7) of k-tile kt into stage kt & 1: 0..3 dy rows 16 p + (2 wave, 2 wave + 1), 4..7 the x rows likewise
  auto stage_piece = [&](int kt, int p) __attribute__((always_inline)) {
    bf16* sY = smem + (kt & 1) * STG;
    if (p < 4) {
      // (the whole offset goes into the lane register: only that one is certain to be range-checked)
      wdma16(rs_y, wlds_addr(sY + (p * 16 + wave * 2) * WT),
             yvo == SENT ? SENT : yvo + (unsigned)((kt * RB + p * 16) * ldy * 2), 0);
    } else {
      const int q = p - 4;
      const int4 ri = *reinterpret_cast<const int4*>(&rinfo[kt & 1][q * 16 + srow]);
      const int trd = taps_packed & 255, trh = (taps_packed >> 8) & 255, trw = taps_packed >> 16;
      const bool ok = colvalid && (unsigned)(ri.y + trd) < (unsigned)g.Di && (unsigned)(ri.z + trh) < (unsigned)g.Hi &&
                      (unsigned)(ri.w + trw) < (unsigned)g.Wi;
      wdma16(rs_x, wlds_addr(sY + TSZ + (q * 16 + wave * 2) * WT), ok ? (unsigned)(ri.x + tapoff) : SENT, 0);
    }
  };

  f32x4 acc[8][4];  // [output-channel tile][column tile]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposing fragment reads: lane (gq, q, p) reads rows 8 gq + q and + 4 (+ 32 for the second half) of a 16-column
  // block; the block swizzle of those rows is the same.  Address = (base | swizzle bits) ^ (block << 5): the lane part
  // sits in one register per operand, the block enters as an immediate XOR.
  const int gq = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  const int r0 = 8 * gq + fq;
  const unsigned ybase = (unsigned)(r0 * WT * 2 + fp * 8) | ((unsigned)wswz(r0) << 5);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  auto frag = [&](unsigned byte) __attribute__((always_inline)) {
    union {
      struct {
        s16x4 a, b;
      } s;
      bf16x8 v;
    } u;
    const char* p = reinterpret_cast<const char*>(smem) + byte;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * WT * 2));
    return u.v;
  };
  bf16x8 yX[8], xX[4], yY[8], xY[4];
  // fragment r of a half (0..3 the wave's column tiles, 4..11 its output-channel tiles) of k-tile stage st, half ks
  auto ldf = [&](bf16x8(&yf)[8], bf16x8(&xf)[4], int st, int ks, int r) __attribute__((always_inline)) {
    const unsigned sb = (unsigned)((st * STG + ks * 32 * WT) * 2);
    if (r < 4) xf[r] = frag(((ybase + sb + TSZ * 2) ^ (unsigned)((wm * 4 + r) << 5)));
    else yf[r - 4] = frag(((ybase + sb) ^ (unsigned)((wn * 8 + (r - 4)) << 5)));
  };

  // ---- prologue: row tables of k-tiles 0 and 1, their requests, the table of k-tile 2; k-tile 0 waited for ----
  if (tid < 2 * RB) decode_rows(tid >> 6, tid & 63);
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 8; ++p) stage_piece(0, p);
#pragma unroll
  for (int p = 0; p < 8; ++p) stage_piece(1, p);
  __syncthreads();
  if (tid < RB) decode_rows(2, tid);
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  if (nkt > 0) {
#pragma unroll
    for (int r = 0; r < 12; ++r) ldf(yX, xX, 0, 0, r);
  }

  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    // ---- first half: products out of X, second half's fragments into Y ----
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int j = n & 3, i = n >> 2;
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yX[i], xX[j], acc[i][j], 0, 0, 0);
      if ((n & 1) == 0 && n < 24) ldf(yY, xY, st, 1, n >> 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // every read of this k-tile's stage is complete, and so are this wave's requests of the next k-tile
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- second half: products out of Y, the next k-tile's first fragments into X, requests of the one after ----
    // (k-tiles past the slice: rows marked outside, dy beyond the descriptor -- zeros land in a stage nobody reads)
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int j = n & 3, i = n >> 2;
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yY[i], xY[j], acc[i][j], 0, 0, 0);
      if ((n & 1) == 0 && n < 24) ldf(yX, xX, st ^ 1, 0, n >> 1);
      if ((n & 3) == 1) stage_piece(kt + 2, n >> 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tid < RB) decode_rows(kt + 3, tid);  // table (kt + 3) & 1 was last read by the requests of k-tile kt + 1
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wn * 128 + i * 16 + (lane >> 4) * 4 + r;
      if (co >= g.Co) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cc = col0 + wm * 64 + j * 16 + (lane & 15);
        if (cc < g.K) atomicAdd(dw + (size_t)co * g.K + cc, acc[i][j][r]);
      }
    }
  }
}

// VC_WGRAD8: 0 never, 1 every geometry the kernel can run, unset: where it measured faster (vc_wgrad8_eligible)
int wgrad8_policy() {
  const char* e = getenv("VC_WGRAD8");
  return e ? (atoi(e) ? 1 : 0) : 2;
}

}  // namespace

bool vc_wgrad8_eligible(const ConvGeom& g, int ldy) {
  const int policy = wgrad8_policy();
  if (!policy) return false;
  if (g.Ci % 8 || g.Co % 8 || g.ldi % 8 || ldy % 8) return false;
  if (g.kd > 255 || g.kh > 255 || g.kw > 255) return false;
  const int64_t xbytes = (int64_t)g.N * g.Di * g.Hi * g.Wi * g.ldi * 2, ybytes = (int64_t)g.M * ldy * 2;
  if (xbytes >= 0x7FFF0000ll || ybytes >= 0x7FFF0000ll) return false;
  if (policy == 1) return true;
  // measured against the 128-wide kernel on the ResNet-50 layer table (512 frames): faster wherever both sides of the
  // tile are at least half full and one is full, except 3 x 3 layers with 128 output channels (4 % slower)
  const int lo = g.Co < g.K ? g.Co : g.K, hi = g.Co < g.K ? g.K : g.Co;
  return lo >= 128 && hi >= 256 && !(g.Co < 192 && g.kd * g.kh * g.kw > 1);
}

// splits: slices of the pixel rows; m_per_split a multiple of 64.  One workgroup per CU and launch where the layer
// allows; never fewer than 512 rows per slice (the fp32 tile a slice adds costs as much as ~100 rows of products).
void vc_wgrad8_plan(const ConvGeom& g, size_t det_capacity, bool det, int* splits, int* m_per_split) {
  const int ntiles = ceil_div(g.Co, WT) * ceil_div(g.K, WT);
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
  *m_per_split = mps;
  *splits = ceil_div(g.M, mps);
}

void vc_wgrad8_launch(const ConvGeom& g, const void* x, const void* dy, int ldy, float* dw, int splits, int m_per_split,
                      size_t split_stride, hipStream_t stream) {
  const int tiles_col = ceil_div(g.K, WT), ntiles = ceil_div(g.Co, WT) * tiles_col;
  hipLaunchKernelGGL(igemm_tn_wgrad8_kernel, dim3(ntiles * splits), dim3(NW8), 0, stream, g, (const bf16*)x,
                     (const bf16*)dy, ldy, dw, m_per_split, ntiles, tiles_col, split_stride);
}
