// Implicit-GEMM NT kernel (forward / data gradient / linear) on 256-pixel x 256-channel output tiles, gfx950.
//
// One workgroup of 8 waves per CU (two waves per SIMD, 256 registers each).  The waves form a 2 x 4 grid, each owning
// 128 pixels x 64 channels of the tile = 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16 (128 registers).
//
// Schedule: software-pipelined and free-running, ONE barrier per 64-deep k-tile.  A k-tile is two 32-deep halves.  While
// a wave multiplies one half out of one fragment set (12 register quads: 8 pixel tiles + 4 channel tiles) it reads the
// other half -- or the first half of the NEXT k-tile -- into a second set, one read after every other product, and
// issues its 8 LDS-DMA requests of the k-tile after next between the products of the second half.  The barrier sits
// between the halves: by then every wave has read both halves of the current k-tile (its 64 KiB stage may be
// overwritten) and has waited for its own requests of the next one.  Two stages of 64 KiB, rows of 128 bytes with the
// 16-byte chunk c of row r at position c ^ ((r >> 1) & 7).
//
// The requests are written as opaque assembly (s_mov m0 / buffer_load_dwordx4 ... offen lds).  With the builtin, and a
// second __shared__ array in the kernel, hipcc puts s_waitcnt vmcnt(0) in front of every LDS read that follows a
// request, which serialises the whole pipeline (430 instead of 840-940 TFLOP/s).
//
// The gather is the buffer-load form of the 128-wide persistent kernel (vc_igemm.hip): per pixel row one 32-bit byte
// offset of its window origin and a bit mask of the filter taps that fall inside the source.  Both are computed once
// per output tile by 256 threads into a table in LDS (two tiles deep; the next tile's rows are decoded while the last
// k-tile of the current one is requested) and turned into per-lane offsets at every tap change.  Lanes whose tap is
// padding (or whose row lies beyond the matrix) carry an offset beyond the descriptor's range, for which the hardware
// writes zeros to LDS; channel rows beyond Co fall outside the weight descriptor in the same way.
//
// Persistent: a workgroup walks output tiles v, v + G, ...; the request stream runs through the tile boundaries, the
// epilogue has an LDS region of its own (2 KiB per wave: the epilogues of the 128-wide kernel, nt_epilogue_lds, called
// once per 64 pixels x 16 channels).
//
// Measured (forward, cold operands, ms / TFLOP/s; 128-wide persistent kernel in brackets): 512 x 7 x 7 x 512 -> 512 3x3
// 0.141 / 839 (0.167 / 711); 2048 -> 512 1x1 0.097 / 541 (0.103 / 511); 4096^3 0.147 / 937 (0.169 / 812); 8192^3
// 1.179 / 933 (1.227 / 896); 512 x 14 x 14 x 256 -> 256 3x3 0.172 / 688 (0.170 / 697); 256 -> 1024 1x1 0.204 / 258
// (0.123 / 427).  It wins where k is deep and the tiles fill the chip in whole rounds, and loses on short k (4-16
// k-tiles per output tile), which is what the automatic choice in vc_nt8_eligible encodes.
#include <stdlib.h>

#include "vc_igemm_dev.h"

namespace {

constexpr int BM8 = 256;   // pixels per tile
constexpr int NT8 = 512;   // threads per workgroup

typedef int i32x4 __attribute__((ext_vector_type(4)));
// One LDS-DMA request written as opaque assembly: the compiler does not see a store to LDS, so it does not put a
// full wait for outstanding requests in front of every later LDS read.  The caller orders requests and reads with
// its own counted waits and barriers.
__device__ __forceinline__ void dma16(const i32x4& rs, unsigned lds_byte, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff)
               : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

template <int MODE, int EPI>
__global__ __launch_bounds__(NT8, 2) void igemm_nt8_kernel(ConvGeom g, const bf16* __restrict__ src,
                                                            const bf16* __restrict__ wgt, const float* __restrict__ bias,
                                                            bf16* __restrict__ out, int relu, int beta, int ntiles,
                                                            float* __restrict__ bn_partial,
                                                            const bf16* __restrict__ tail_y,
                                                            const uint8_t* __restrict__ tail_bits) {
  constexpr int TN = 256;
  constexpr int ASZ = BM8 * BK, WSZ = TN * BK, STG = ASZ + WSZ;  // elements per stage
  constexpr int EPW = 1024;                                       // epilogue image per wave: 64 pixels x 16 channels
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * STG + 8 * EPW];
  // per row of an output tile: byte offset of its window origin in the source, and its tap mask; two tiles deep
  __shared__ uint2 rowtab[2][BM8];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 2, wc = wave & 3;
  const int tiles_n = (g.Co + TN - 1) / TN;
  int vblock;
  xcd_remap(vblock);
  const int G = gridDim.x;
  const int lrow = tid >> 3, pchunk = tid & 7;
  const int lchunk = pchunk ^ ((lrow >> 1) & 7);
  const int ncb = (g.Ci + BK - 1) / BK;
  const int ntaps = g.kd * g.kh * g.kw;
  const int nk = ntaps * ncb;
  const bool ktail = (g.Ci % BK) != 0;

  struct KCursor {
    int tile, kt, tap, cb, rd, rh, rw, st, par;
  };
  auto advance = [&](KCursor& c) {  // tap-major inside an output tile (see the ablation notes: channel-block-major is slower)
    if (++c.cb == ncb) {
      c.cb = 0;
      ++c.tap;
      if (++c.rw == g.kw) {
        c.rw = 0;
        if (++c.rh == g.kh) {
          c.rh = 0;
          ++c.rd;
        }
      }
    }
    if (++c.kt == nk) {
      c.kt = 0;
      c.tap = c.cb = c.rd = c.rh = c.rw = 0;
      c.tile += G;
      c.par ^= 1;
    }
    c.st ^= 1;
  };
  KCursor cu{vblock, 0, 0, 0, 0, 0, 0, 0, 0};

  constexpr unsigned SENT = 0xFFFFFFF0u;
  unsigned avo[4], wvo0 = 0;
  int s_kb = 0, s_wk = 0;
  bool s_dead = false;
  const i32x4 rs_a = {(int)(uintptr_t)src, (int)((uintptr_t)src >> 32),
                      (int)((unsigned)(g.N * g.Di * g.Hi * g.Wi) * (unsigned)g.ldi * 2u), 0x00020000};
  const i32x4 rs_w = {(int)(uintptr_t)wgt, (int)((uintptr_t)wgt >> 32), (int)((unsigned)g.Co * (unsigned)g.ldw * 2u),
                      0x00020000};

  // state of the cursor's k-tile (rows of a new output tile, offsets of a new tap): before its first request
  // rows of output tile `tile` into table `par` (one row per thread of the first four waves)
  auto write_rows = [&](int tile, int par) {
    if (tid < BM8 && tile < ntiles) {
      const RowPos r = decode_row<MODE>(g, (tile / tiles_n) * BM8 + tid);
      rowtab[par][tid] = make_uint2((unsigned)((((r.n * g.Di + r.d0) * g.Hi + r.h0) * g.Wi + r.w0) * g.ldi * 2),
                                    tap_mask<MODE>(g, r));
    }
  };
  // state of the cursor's k-tile (offsets of a new tap, channel row of a new output tile): before its first request.
  // The rows of the tile after the cursor's are decoded while its last k-tile is requested: a barrier lies between
  // that and the first use.
  auto stage_prepare = [&]() {
    if (cu.kt == 0) {
      const int n0 = (cu.tile % tiles_n) * TN;
      wvo0 = (unsigned)((n0 + (lrow >> 5) * 64 + (lrow & 31)) * g.ldw * 2 + lchunk * 16);
    }
    if (cu.cb == 0) {
      int toff = ((cu.rd * g.Hi + cu.rh) * g.Wi + cu.rw) * g.ldi * 2;
      if (MODE != 0) toff = -toff;
      toff += lchunk * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // i = h * 2 + gg
        const uint2 e = rowtab[cu.par][(i & 1) * 128 + (i >> 1) * 64 + lrow];
        avo[i] = ((e.y >> cu.tap) & 1u) ? e.x + (unsigned)toff : SENT;
      }
      s_wk = MODE == 2 ? ((cu.rd * g.wt_kh + g.wt_h0 + g.wt_hs * cu.rh) * g.wt_kw + g.wt_w0 + g.wt_ws * cu.rw) * g.Ci * 2
                       : cu.tap * g.Ci * 2;
    }
    s_kb = cu.cb * (BK * 2);
    s_dead = ktail && cu.cb * BK + lchunk * 8 >= g.Ci;
    if (cu.kt == nk - 1) write_rows(cu.tile + G, cu.par ^ 1);
  };
  // request p (0..7) of the cursor's k-tile: 0,1 pixel rows [0,64) of both wave rows; 2,3 / 4,5 the first / second 32
  // channels of every wave column; 6,7 pixel rows [64,128); the last one moves the cursor on
  auto stage_piece = [&](int p, bool live) {
    bf16* sA = smem + cu.st * STG;
    bf16* sW = sA + ASZ;
    const int gg = p & 1;
    if (p < 2)
      dma16(rs_a, lds_addr(sA + (gg * 128 + 8 * wave) * BK), s_dead || !live ? SENT : avo[gg], s_kb);
    else if (p < 6) {
      const int c = (p - 2) >> 1;
      dma16(rs_w, lds_addr(sW + ((gg * 2 + (wave >> 2)) * 64 + c * 32 + (wave & 3) * 8) * BK),
            s_dead || !live ? SENT : wvo0 + (unsigned)((gg * 128 + c * 32) * g.ldw * 2), s_wk + s_kb);
    } else {
      dma16(rs_a, lds_addr(sA + (gg * 128 + 64 + 8 * wave) * BK), s_dead || !live ? SENT : avo[2 + gg], s_kb);
      if (p == 7 && live) advance(cu);
    }
  };

  f32x4 acc[2][4][4];  // [pixel half][channel tile][pixel tile]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, q = lane >> 4, x = (l15 >> 1) & 7;
  const int c0 = ((q ^ x) << 3), c1 = (((q ^ x) ^ 4) << 3);
  const int aoff = (wp * 128 + l15) * BK, woff = ASZ + (wc * 64 + l15) * BK;

  bf16x8 afX[8], wfX[4], afY[8], wfY[4];
  // fragment r of a half: 0..3 the wave's channel tiles, 4..11 its pixel tiles
  auto ldX = [&](const bf16* tS, int r, int cc) {
    if (r < 4) wfX[r] = *reinterpret_cast<const bf16x8*>(tS + woff + cc + r * 16 * BK);
    else afX[r - 4] = *reinterpret_cast<const bf16x8*>(tS + aoff + cc + (r - 4) * 16 * BK);
  };
  auto ldY = [&](const bf16* tS, int r, int cc) {
    if (r < 4) wfY[r] = *reinterpret_cast<const bf16x8*>(tS + woff + cc + r * 16 * BK);
    else afY[r - 4] = *reinterpret_cast<const bf16x8*>(tS + aoff + cc + (r - 4) * 16 * BK);
  };

  // ---- prologue: k-tiles 0 and 1 requested, k-tile 0 waited for, its first half in registers ----
  bool two = false;
  write_rows(cu.tile, 0);
  __syncthreads();
  if (cu.tile < ntiles) {
    stage_prepare();
#pragma unroll
    for (int p = 0; p < 8; ++p) stage_piece(p, true);
  }
  __syncthreads();
  if (cu.tile < ntiles) {
    stage_prepare();
#pragma unroll
    for (int p = 0; p < 8; ++p) stage_piece(p, true);
    two = true;
  }
  if (two) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  int c_tile = vblock, c_kt = 0, cs = 0;
  if (c_tile < ntiles) {
#pragma unroll
    for (int r = 0; r < 12; ++r) ldX(smem, r, c0);
  }

  while (c_tile < ntiles) {
    const bool last = c_kt == nk - 1;
    const bf16* tS = smem + cs * STG;
    const bf16* tN = smem + (cs ^ 1) * STG;
    // ---- first half: products out of X, second half's fragments into Y ----
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int ci = n & 3, jj = n >> 2;
      acc[jj >> 2][ci][jj & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfX[ci], afX[jj], acc[jj >> 2][ci][jj & 3], 0, 0, 0);
      if ((n & 1) == 0 && n < 24) ldY(tS, n >> 1, c1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // every read of this k-tile's stage is complete, and so are this wave's requests of the next k-tile
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // is there a k-tile after this one (for this workgroup)?  its first half goes into X
    const bool more = cu.tile < ntiles;
    if (more) stage_prepare();
    // ---- second half: products out of Y, the next k-tile's first fragments into X, requests of the one after ----
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int ci = n & 3, jj = n >> 2;
      acc[jj >> 2][ci][jj & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfY[ci], afY[jj], acc[jj >> 2][ci][jj & 3], 0, 0, 0);
      if ((n & 1) == 0 && n < 24) ldX(tN, n >> 1, c0);  // past the last k-tile: stale LDS, never multiplied
      if ((n & 3) == 1) stage_piece(n >> 2, more);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (last) {
      const int tile_m = c_tile / tiles_n, tile_n = c_tile - tile_m * tiles_n;
      bf16* img = smem + 2 * STG + wave * EPW;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          const int m0 = tile_m * BM8 + wp * 128 + h * 64, n0 = tile_n * TN + wc * 64 + ci * 16;
          f32x4(&a1)[1][4] = *reinterpret_cast<f32x4(*)[1][4]>(&acc[h][ci]);
          if (bias || (EPI != EPI_NONE && EPI != EPI_BN_TAIL))
            nt_epilogue_lds<MODE, 1, EPI, true>(g, a1, m0, n0, 0, 0, lane, bias, out, relu, beta, img, bn_partial, tail_y,
                                                tail_bits);
          else
            nt_epilogue_lds<MODE, 1, EPI, false>(g, a1, m0, n0, 0, 0, lane, nullptr, out, relu, beta, img, bn_partial,
                                                 tail_y, tail_bits);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[h][ci][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      c_kt = 0;
      c_tile += G;
    } else {
      ++c_kt;
    }
    cs ^= 1;
  }
}

// VC_NT8: 0 never, 1 every geometry the kernel can run, unset: where it measured faster (read at every call so that one
// process can compare both forms).
int nt8_policy() {
  const char* e = getenv("VC_NT8");
  return e ? (atoi(e) ? 1 : 0) : 2;
}

template <int MODE, int EPI>
void nt8_go(const ConvGeom& g, int grid, int tiles, hipStream_t stream, const void* src, const void* wgt, const float* bias,
            void* out, int relu, int beta, float* bn_partial, const void* tail_y, const uint8_t* tail_bits) {
  hipLaunchKernelGGL((igemm_nt8_kernel<MODE, EPI>), dim3(grid), dim3(NT8), 0, stream, g, (const bf16*)src,
                     (const bf16*)wgt, bias, (bf16*)out, relu, beta, tiles, bn_partial, (const bf16*)tail_y, tail_bits);
}

}  // namespace

bool vc_nt8_eligible(const ConvGeom& g, int mode, int out_f32, int epi) {
  const int policy = nt8_policy();
  if (!policy || out_f32 || g.ksplit != 1) return false;
  if (!((mode == 0 && epi == EPI_NONE) ||
        (mode == 1 && (epi == EPI_NONE || epi == EPI_BN_PRE || epi == EPI_BN_TAIL))))
    return false;
  const int taps = g.kd * g.kh * g.kw;
  if (taps > 32 || g.ldi % 8 != 0 || g.K < 8) return false;
  if (!(g.Ci % BK == 0 || (taps == 1 && g.Ci % 8 == 0))) return false;
  if (mode == 1 && (g.sd != 1 || g.sh != 1 || g.sw != 1)) return false;
  const int64_t abytes = (int64_t)g.N * g.Di * g.Hi * g.Wi * g.ldi * 2, wbytes = (int64_t)g.Co * g.ldw * 2;
  if (abytes >= 0xFFFF0000ll || wbytes >= 0xFFFF0000ll) return false;
  if (g.Co < 192) return false;
  if (policy == 1) return true;
  // where it measured faster than the 128-wide kernel: at least 32 k-tiles per output tile, and a tile count that fills
  // the chip in whole rounds (one partial round, an exact multiple, or many rounds)
  const int nk = taps * ceil_div(g.Ci, BK);
  const int tiles = ceil_div(g.M, BM8) * ceil_div(g.Co, 256), cus = vc_device_cus();
  // (a single round must fill at least three quarters of the chip: 75 tiles of a 7 x 7 Swin layer ran 0.101 ms against
  // 0.061 ms on the 128-wide tiles)
  return nk >= 32 && ((tiles <= cus && 4 * tiles >= 3 * cus) || tiles % cus == 0 || tiles >= 3 * cus);
}

void vc_nt8_launch(const ConvGeom& g, int mode, int epi, hipStream_t stream, const void* src, const void* wgt,
                   const float* bias, void* out, int relu, int beta, float* bn_partial, const void* tail_y,
                   const uint8_t* tail_bits) {
  const int tiles = ceil_div(g.M, BM8) * ceil_div(g.Co, 256);
  const int cus = vc_device_cus();
  const int grid = tiles < cus ? tiles : cus;
  if (mode == 0) nt8_go<0, EPI_NONE>(g, grid, tiles, stream, src, wgt, bias, out, relu, beta, bn_partial, tail_y, tail_bits);
  else if (epi == EPI_BN_PRE)
    nt8_go<1, EPI_BN_PRE>(g, grid, tiles, stream, src, wgt, bias, out, relu, beta, bn_partial, tail_y, tail_bits);
  else if (epi == EPI_BN_TAIL)
    nt8_go<1, EPI_BN_TAIL>(g, grid, tiles, stream, src, wgt, bias, out, relu, beta, bn_partial, tail_y, tail_bits);
  else nt8_go<1, EPI_NONE>(g, grid, tiles, stream, src, wgt, bias, out, relu, beta, bn_partial, tail_y, tail_bits);
}
