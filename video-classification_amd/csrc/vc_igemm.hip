// Implicit-GEMM convolution / linear kernels for gfx950 (MI355X).
//
// One kernel family serves every dense contraction on the path: Conv2d, Conv3d and
// Linear (a 1x1x1 convolution over a 1x1x1 image), forward, data-gradient and
// weight-gradient.  Activations are NDHWC (channels innermost), weights are packed
// [Cout][kd*kh*kw*Cin] with the tap index outside the channel index, so that every
// 8-channel group is one 16-byte load.
//
//   * igemm_nt: out[pixel][co] = sum_k gather(src)[pixel][k] * wgt[co][k]
//       MODE 0: forward gather (in = out*stride - pad + tap)
//       MODE 1: data-gradient gather (in = (out + pad - tap)/stride when divisible)
//       MODE 2: one output phase of a strided data gradient: the rows are the dx pixels with
//               (h, w) = phase (mod stride) and only the taps that reach them are in the
//               reduction (a stride-1 MODE 1 gather over a sub-sampled kernel), so no
//               product is spent on a tap that fails the divisibility test
//     128x128x64 tiles, 4 waves (2x2), 4x4 v_mfma_f32_16x16x32_bf16 tiles per wave,
//     register-staged global->LDS double buffering, XOR-swizzled 128-byte LDS rows.
//   * igemm_tn (weight gradient): dw[co][k] += sum_pixel dy[pixel][co] * gather(x)[pixel][k]
//     Both operands are pixel-major in memory, i.e. transposed with respect to what an
//     MFMA fragment wants, so the LDS tiles stay pixel-major and the fragments are read
//     with ds_read_b64_tr_b16 (the hardware transposing LDS read of gfx950).
//     Split over the pixel dimension; partial tiles are combined with fp32 atomics.
//   * naive kernels (any element type): one thread per output element, fp32 accumulate.
//     They are the dtype = fp32 implementation (exact-order-independent fp32 reference
//     grade) and the on-device cross-check for the MFMA kernels.
#include <stdlib.h>

#include "vc_igemm_dev.h"

namespace {


// GLDS = true : tiles are staged global -> LDS by LDS-DMA (no VGPR round trip, no ds_write); the XOR
//               swizzle of the LDS image is applied to each lane's SOURCE address.
// GLDS = false: register-staged variant (global_load -> VGPR -> ds_write), kept for A/B measurements.
// TN_ = 128, or 64 for layers with at most 64 output channels (first layers: 3 -> 32 / 64, 1 -> 32; 32 -> 64).
template <int MODE, bool GLDS, int TN_ = 128>
__global__ __launch_bounds__(NTHR, 2) void igemm_nt_bf16_kernel(ConvGeom g, const bf16* __restrict__ src,
                                                                const bf16* __restrict__ wgt,
                                                                const float* __restrict__ bias, void* __restrict__ out,
                                                                int relu, int out_f32, int beta) {
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * (BM + TN_) * BK];
  constexpr int WI = TN_ / 32;     // 16-channel tiles per wave
  constexpr int NWROW = TN_ / 32;  // weight rows staged per thread

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.Co + TN_ - 1) / TN_;
  int logical;
  xcd_remap(logical);
  const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * TN_;

  // staging role of this thread: LDS row lrow (+32*i), physical 16-byte slot pchunk of the row,
  // which holds logical k-chunk lchunk (same XOR for the four rows: 32*i does not change (row>>1)&7)
  const int lrow = tid >> 3, pchunk = tid & 7;
  const int lchunk = pchunk ^ ((lrow >> 1) & 7);
  RowPos rp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rp[i] = decode_row<MODE>(g, m0 + lrow + 32 * i);

  const int nk = (g.K + BK - 1) / BK;
  bf16x8 areg[4], wreg[NWROW];

  auto stage_dma = [&](int kt, int buf) {
    const int kc = kt * (BK / 8) + lchunk;
    const bool kvalid = kc * 8 < g.K;
    int rd, rh, rw, c;
    decode_chunk(g, kc, rd, rh, rw, c);
    const int wcol = weight_col<MODE>(g, kc, rd, rh, rw, c);
    bf16* sA = smem + buf * (BM + TN_) * BK;
    bf16* sW = sA + BM * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const void* asrc = vc_zero_src;
      const int pix = gather_pixel<MODE>(g, rp[i], rd, rh, rw);
      if (kvalid && pix >= 0) asrc = src + (size_t)pix * g.ldi + c;
      // this wave's instruction fills rows [32*i + 8*wave, +8): 8 rows x 128 B = 64 lanes x 16 B
      glds16(asrc, sA + (32 * i + 8 * wave) * BK);
    }
#pragma unroll
    for (int i = 0; i < NWROW; ++i) {
      const void* wsrc = vc_zero_src;
      const int co = n0 + lrow + 32 * i;
      if (kvalid && co < g.Co) wsrc = wgt + (size_t)co * g.ldw + wcol;
      glds16(wsrc, sW + (32 * i + 8 * wave) * BK);
    }
  };
  auto load_tile = [&](int kt) {
    const int kc = kt * (BK / 8) + lchunk;
    const bool kvalid = kc * 8 < g.K;
    int rd, rh, rw, c;
    decode_chunk(g, kc, rd, rh, rw, c);
    const int wcol = weight_col<MODE>(g, kc, rd, rh, rw, c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      const int pix = gather_pixel<MODE>(g, rp[i], rd, rh, rw);
      if (kvalid && pix >= 0) v = *reinterpret_cast<const bf16x8*>(src + (size_t)pix * g.ldi + c);
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NWROW; ++i) {
      bf16x8 w = {0, 0, 0, 0, 0, 0, 0, 0};
      const int co = n0 + lrow + 32 * i;
      if (kvalid && co < g.Co) w = *reinterpret_cast<const bf16x8*>(wgt + (size_t)co * g.ldw + wcol);
      wreg[i] = w;
    }
  };
  auto store_tile = [&](int buf) {
    bf16* sA = smem + buf * (BM + TN_) * BK;
    bf16* sW = sA + BM * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<bf16x8*>(sA + (lrow + 32 * i) * BK + pchunk * 8) = areg[i];
#pragma unroll
    for (int i = 0; i < NWROW; ++i) *reinterpret_cast<bf16x8*>(sW + (lrow + 32 * i) * BK + pchunk * 8) = wreg[i];
  };

  f32x4 acc[WI][4];
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (GLDS) {
    stage_dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      if (GLDS) stage_dma(kt + 1, buf ^ 1);
      else load_tile(kt + 1);
    }
    const bf16* sA = smem + buf * (BM + TN_) * BK;
    const bf16* sW = sA + BM * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = ks * 4 + (lane >> 4);
      bf16x8 wf[WI], af[4];
#pragma unroll
      for (int i = 0; i < WI; ++i)
        wf[i] = *reinterpret_cast<const bf16x8*>(sW + nt_lds_off(wn * (WI * 16) + i * 16 + (lane & 15), chunk));
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(sA + nt_lds_off(wm * 64 + i * 16 + (lane & 15), chunk));
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    if (GLDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (kt + 1 < nk) {
      store_tile(buf ^ 1);
    }
    __syncthreads();
  }

  nt_epilogue<MODE, WI>(g, acc, m0, n0, wm, wn, lane, bias, out, relu, out_f32, beta);
}

// NT kernel for source channel counts that are a multiple of BK (every layer past a network's first): a
// k-tile then lies inside ONE tap, so the tap decode and the bounds tests run once per tap (on uniform
// values where possible) and the per-tile address work is one pointer increment per staged row.
template <int MODE>
__global__ __launch_bounds__(NTHR, 2) void igemm_nt_fast_kernel(ConvGeom g, const bf16* __restrict__ src,
                                                                const bf16* __restrict__ wgt,
                                                                const float* __restrict__ bias, void* __restrict__ out,
                                                                int relu, int out_f32, int beta) {
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * (BM + BN) * BK];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.Co + BN - 1) / BN;
  int logical;
  xcd_remap(logical);
  const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int lrow = tid >> 3, pchunk = tid & 7;
  const int lchunk = pchunk ^ ((lrow >> 1) & 7);
  RowPos rp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rp[i] = decode_row<MODE>(g, m0 + lrow + 32 * i);

  const int ncb = g.Ci / BK;  // k-tiles per tap
  const int nk = g.K / BK;
  const bf16* zero = reinterpret_cast<const bf16*>(vc_zero_src);

  // staged-row pointers: advance by BK elements per k-tile inside a tap; rows that read padding (or lie
  // beyond the matrix) point at the zero page and do not advance
  const bf16* ap[4];
  const bf16* wp[4];
  int ainc[4], winc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = n0 + lrow + 32 * i;
    const bool ok = co < g.Co;
    wp[i] = ok ? wgt + (size_t)co * g.ldw + lchunk * 8 : zero;
    winc[i] = ok ? BK : 0;
    ap[i] = zero;
    ainc[i] = 0;
  }
  int tap = 0, cb = 0;  // uniform over the block

  auto stage = [&](int buf) {
    if (cb == 0) {
      int t2, rd, rh, rw;
      fd_divmod(tap, g.fkw, t2, rw);
      fd_divmod(t2, g.fkh, rd, rh);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pix = gather_pixel<MODE>(g, rp[i], rd, rh, rw);
        const bool ok = pix >= 0;
        ap[i] = ok ? src + (size_t)pix * g.ldi + lchunk * 8 : zero;
        ainc[i] = ok ? BK : 0;
      }
      if (MODE == 2) {  // the taps of a phase are not adjacent in the packed weight row
        const int wcol = ((rd * g.wt_kh + g.wt_h0 + g.wt_hs * rh) * g.wt_kw + g.wt_w0 + g.wt_ws * rw) * g.Ci;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = n0 + lrow + 32 * i;
          if (co < g.Co) wp[i] = wgt + (size_t)co * g.ldw + wcol + lchunk * 8;
        }
      }
    }
    bf16* sA = smem + buf * (BM + BN) * BK;
    bf16* sW = sA + BM * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(ap[i], sA + (32 * i + 8 * wave) * BK);
      glds16(wp[i], sW + (32 * i + 8 * wave) * BK);
      ap[i] += ainc[i];
      wp[i] += winc[i];
    }
    if (++cb == ncb) {
      cb = 0;
      ++tap;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row r of a tile keeps its 16-byte chunks XORed with (r >> 1) & 7, which for the
  // rows one lane reads (l15 + 16*i) is the same value for every i
  const int l15 = lane & 15, q = lane >> 4, x = (l15 >> 1) & 7;
  const int c0 = ((q ^ x) << 3), c1 = (((q ^ x) ^ 4) << 3);
  const int aoff = (wm * 64 + l15) * BK, woff = BM * BK + (wn * 64 + l15) * BK;

  auto compute = [&](const bf16* tile) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cc = ks ? c1 : c0;
      bf16x8 wf[4], af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wf[i] = *reinterpret_cast<const bf16x8*>(tile + woff + cc + i * 16 * BK);
        af[i] = *reinterpret_cast<const bf16x8*>(tile + aoff + cc + i * 16 * BK);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
  };

  if (nk > 0) stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 1 < nk) stage(1);
    compute(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 2 < nk) stage(0);
    compute(smem + (BM + BN) * BK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  nt_epilogue<MODE>(g, acc, m0, n0, wm, wn, lane, bias, out, relu, out_f32, beta);
}


// Persistent form of the fast NT kernel: a block walks output tiles v, v + G, v + 2G, ... and its staging
// pipeline runs straight through the tile boundaries, so the first k-tile of the next output tile is already
// streaming in while the epilogue of the current one is written.  Layers with a short reduction (1x1
// convolutions with 64..256 channels are 1..4 k-tiles deep) otherwise pay a full memory latency per block.
//
// BL = true stages with buffer LDS-DMA loads (buffer_load_dwordx4 ... lds) instead of global ones.  A lane whose
// offset lies beyond the descriptor's range makes the hardware write ZEROS to its 16 bytes of LDS (measured; the
// range check looks at the vector offset only, not at the scalar one), so padding taps, rows beyond the matrix
// and weight rows beyond Co are a sentinel offset: no zero page, no 64-bit pointers, no per-row increments.
// A staged row is one 32-bit byte offset of its window origin plus a bit mask of the taps that fall inside the
// source, both computed once per output tile; per filter tap a row costs an add and a select, inside a tap
// nothing at all (the k-tile's advance is the instruction's scalar offset).  Needs at most 32 filter taps, a
// stride-1 gather in MODE 1 and operands of less than 4 GiB.
// BM_ = 256 (with TN_ = 64): the four waves stack along the pixel rows, each with all 64 channels of a 64-row slab,
// instead of two by two with 32 channels each: a wave's epilogue then writes whole 128-byte rows of a 64-channel tensor.
template <int MODE, int TN_, int EPI = EPI_NONE, bool BL = false, int NA = 3, int BM_ = 128>
__global__ __launch_bounds__(NTHR, 2) void igemm_nt_persist_kernel(ConvGeom g, const bf16* __restrict__ src,
                                                                   const bf16* __restrict__ wgt,
                                                                   const float* __restrict__ bias,
                                                                   void* __restrict__ out, int relu, int out_f32,
                                                                   int beta, int ntiles,
                                                                   float* __restrict__ bn_partial,
                                                                   const bf16* __restrict__ tail_y = nullptr,
                                                                   const uint8_t* __restrict__ tail_bits = nullptr) {
  // NA staged k-tiles of the source rows and two of the weight rows.  NA = 3 keeps one source k-tile in flight ACROSS
  // the barrier (a counted vmcnt leaves its loads outstanding): the source streams from HBM, whose latency under load
  // is several times one k-tile of products; the weights come out of L2 and get by with one k-tile of lead.
  constexpr bool TALL = BM_ == 256;
  static_assert(BM_ == 128 || (TALL && TN_ == 64 && NA == 2), "tile rows");
  constexpr int NAR = BM_ / 32;                  // source rows staged per thread
  constexpr int ASZ = BM_ * BK, WSZ = TN_ * BK;  // elements per staged k-tile
  __shared__ __attribute__((aligned(1024))) bf16 smem[NA * ASZ + 2 * WSZ];
  bf16* const sA0 = smem;
  bf16* const sW0 = smem + NA * ASZ;
  constexpr int WI = TALL ? TN_ / 16 : TN_ / 32;   // 16-channel tiles per wave
  constexpr int NWROW = TN_ / 32;  // weight rows staged per thread

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = TALL ? wave : wave >> 1, wn = TALL ? 0 : wave & 1;
  const int tiles_n = (g.Co + TN_ - 1) / TN_;
  int vblock;
  xcd_remap(vblock);
  const int G = gridDim.x;

  const int lrow = tid >> 3, pchunk = tid & 7;
  const int lchunk = pchunk ^ ((lrow >> 1) & 7);
  // k-tiles per tap; a single-tap layer (linear, 1x1 convolution) may have any channel count that is a
  // multiple of 8: the chunks of its last k-tile that lie beyond Ci read zeros
  // > 1: single tap, Ci % (BK * KS) == 0, fp32 partial outputs (plain forward layers only: the epilogue variants are
  // at their register limit and never run split)
  const int KS = (EPI == EPI_NONE && MODE == 0 && !BL) ? g.ksplit : 1;
  const int ncb = (g.Ci + BK - 1) / BK / KS;     // k-tiles per tap (and slice)
  const int nk = g.kd * g.kh * g.kw * ncb;       // k-tiles per work unit (>= 1)
  const bool ktail = (g.Ci % BK) != 0;
  const bf16* zero = reinterpret_cast<const bf16*>(vc_zero_src);
  // work unit -> output tile and the element offset of its slice inside a source / weight row
  auto unit_tile = [&](int v) { return KS > 1 ? v / KS : v; };
  auto unit_koff = [&](int v) { return KS > 1 ? (v - (v / KS) * KS) * ncb * BK : 0; };

  // A staging cursor: which k-tile of which output tile comes next.  The source rows and the weight rows have one
  // each (the source runs further ahead); both are wave-uniform.
  struct KCursor {
    int tile, kt, tap, cb, rd, rh, rw;
  };
  auto advance = [&](KCursor& c) {
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
    }
  };
  KCursor ca{vblock, 0, 0, 0, 0, 0, 0}, cw = ca;

  // ---- staging by global LDS-DMA loads ----
  RowPos rp[NAR];
  const bf16* ap[NAR];
  const bf16* wp[4];
  int ainc[NAR], winc[4];

  auto stage_a_gl = [&](bf16* sA, KCursor& c) {
    if (c.kt == 0) {  // first k-tile of an output tile: its rows
      const int m0 = (unit_tile(c.tile) / tiles_n) * BM_;
#pragma unroll
      for (int i = 0; i < NAR; ++i) rp[i] = decode_row<MODE>(g, m0 + lrow + 32 * i);
    }
    if (c.cb == 0) {
      const int koff = unit_koff(c.tile);
#pragma unroll
      for (int i = 0; i < NAR; ++i) {
        const int pix = gather_pixel<MODE>(g, rp[i], c.rd, c.rh, c.rw);
        const bool ok = pix >= 0;
#ifdef VC_ABL_HOTA  // ablation build: source rows from a 64 KiB window (time only)
        ap[i] = ok ? src + (((size_t)pix * g.ldi) & 0x7FFF) + lchunk * 8 + koff : zero;
#else
        ap[i] = ok ? src + (size_t)pix * g.ldi + lchunk * 8 + koff : zero;
#endif
        ainc[i] = ok ? BK : 0;
      }
    }
    const bool dead = ktail && c.cb * BK + lchunk * 8 >= g.Ci;  // this thread's chunk of a partial last k-tile
#pragma unroll
    for (int i = 0; i < NAR; ++i) glds16(dead ? zero : ap[i], sA + (32 * i + 8 * wave) * BK);
#pragma unroll
    for (int i = 0; i < NAR; ++i) ap[i] += ainc[i];
    advance(c);
  };
  auto stage_w_gl = [&](bf16* sW, KCursor& c) {
    const int rt = unit_tile(c.tile);
    const int tile_m = rt / tiles_n, n0 = (rt - tile_m * tiles_n) * TN_;
    if (c.kt == 0) {
      const int koff = unit_koff(c.tile);
#pragma unroll
      for (int i = 0; i < NWROW; ++i) {
        const int co = n0 + lrow + 32 * i;
        const bool ok = co < g.Co;
        wp[i] = ok ? wgt + (size_t)co * g.ldw + lchunk * 8 + koff : zero;
        winc[i] = ok ? BK : 0;
      }
    }
    if (MODE == 2 && c.cb == 0) {
      const int wcol = ((c.rd * g.wt_kh + g.wt_h0 + g.wt_hs * c.rh) * g.wt_kw + g.wt_w0 + g.wt_ws * c.rw) * g.Ci;
#pragma unroll
      for (int i = 0; i < NWROW; ++i) {
        const int co = n0 + lrow + 32 * i;
        if (co < g.Co) wp[i] = wgt + (size_t)co * g.ldw + wcol + lchunk * 8;
      }
    }
    const bool dead = ktail && c.cb * BK + lchunk * 8 >= g.Ci;
#pragma unroll
    for (int i = 0; i < NWROW; ++i) glds16(dead ? zero : wp[i], sW + (32 * i + 8 * wave) * BK);
#pragma unroll
    for (int i = 0; i < NWROW; ++i) wp[i] += winc[i];
    advance(c);
  };

  // ---- staging by buffer LDS-DMA loads (see the comment above the kernel) ----
  constexpr unsigned SENT = 0xFFFFFFF0u;  // beyond every descriptor's range
  unsigned abase[NAR], amask[NAR], avo[NAR], wvo[NWROW];
  int s_wk = 0;  // uniform: byte offset of the current tap's weights in a packed row
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(src), 0, (int)((unsigned)(g.N * g.Di * g.Hi * g.Wi) * (unsigned)g.ldi * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(wgt), 0, (int)((unsigned)g.Co * (unsigned)g.ldw * 2u), 0x00020000);

  auto stage_a_bl = [&](bf16* sA, KCursor& c) {
    if (c.kt == 0) {
      const int m0 = (c.tile / tiles_n) * BM_;
#pragma unroll
      for (int i = 0; i < NAR; ++i) {
        const RowPos r = decode_row<MODE>(g, m0 + lrow + 32 * i);  // rows beyond M: coordinates far outside
        abase[i] = (unsigned)((((r.n * g.Di + r.d0) * g.Hi + r.h0) * g.Wi + r.w0) * g.ldi * 2 + lchunk * 16);
        amask[i] = tap_mask<MODE>(g, r);
      }
    }
    if (c.cb == 0) {
      int toff = ((c.rd * g.Hi + c.rh) * g.Wi + c.rw) * g.ldi * 2;
      if (MODE != 0) toff = -toff;
#pragma unroll
      for (int i = 0; i < NAR; ++i) avo[i] = ((amask[i] >> c.tap) & 1u) ? abase[i] + (unsigned)toff : SENT;
    }
    const int kb = c.cb * (BK * 2);  // byte advance inside the tap: the instructions' scalar offset
    const bool dead = ktail && c.cb * BK + lchunk * 8 >= g.Ci;
#pragma unroll
    for (int i = 0; i < NAR; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(sA + (32 * i + 8 * wave) * BK), 16,
                                               (int)(dead ? SENT : avo[i]), kb, 0, 0);
    advance(c);
  };
  auto stage_w_bl = [&](bf16* sW, KCursor& c) {
    if (c.kt == 0) {
      const int tile_m = c.tile / tiles_n, n0 = (c.tile - tile_m * tiles_n) * TN_;
#pragma unroll
      for (int i = 0; i < NWROW; ++i) {
        const int co = n0 + lrow + 32 * i;
        wvo[i] = co < g.Co ? (unsigned)(co * g.ldw * 2 + lchunk * 16) : SENT;
      }
    }
    if (c.cb == 0)
      s_wk = MODE == 2 ? ((c.rd * g.wt_kh + g.wt_h0 + g.wt_hs * c.rh) * g.wt_kw + g.wt_w0 + g.wt_ws * c.rw) * g.Ci * 2
                       : c.tap * g.Ci * 2;
    const int kb = c.cb * (BK * 2);
    const bool dead = ktail && c.cb * BK + lchunk * 8 >= g.Ci;
#pragma unroll
    for (int i = 0; i < NWROW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(sW + (32 * i + 8 * wave) * BK), 16,
                                               (int)(dead ? SENT : wvo[i]), s_wk + kb, 0, 0);
    advance(c);
  };
#define VC_STAGE_A(buf)                   \
  do {                                    \
    if constexpr (BL) stage_a_bl(buf, ca); \
    else stage_a_gl(buf, ca);             \
  } while (0)
#define VC_STAGE_W(buf)                   \
  do {                                    \
    if constexpr (BL) stage_w_bl(buf, cw); \
    else stage_w_gl(buf, cw);             \
  } while (0)

  f32x4 acc[WI][4];
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, q = lane >> 4, x = (l15 >> 1) & 7;
  const int c0 = ((q ^ x) << 3), c1 = (((q ^ x) ^ 4) << 3);
  const int aoff = (wm * 64 + l15) * BK, woff = (wn * (WI * 16) + l15) * BK;

  auto compute = [&](const bf16* tA, const bf16* tW) {
#ifdef VC_ABL_BNPRO  // ablation build (time only): relu(y * s + t) on the staged source k-tile, in place, before the
    {                 // products -- what a BatchNorm apply folded into its consumer's operand path would cost
      bf16* tAw = const_cast<bf16*>(tA);
#pragma unroll
      for (int i = 0; i < NAR; ++i) {
        bf16x8* pv = reinterpret_cast<bf16x8*>(tAw + (32 * i + lrow) * BK + pchunk * 8);
        bf16x8 v = *pv;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaxf(fmaf((float)v[e], 1.00390625f, 0.0009765625f * (float)(lchunk + e)), 0.f);
        *pv = v;
      }
      lds_barrier();
    }
#endif
    // all fragments of the k-tile are requested before the first product, so that one LDS round trip is exposed
    // per k-tile instead of one per group of products
    bf16x8 wf[2][WI], af[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cc = ks ? c1 : c0;
#pragma unroll
      for (int i = 0; i < WI; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8*>(tW + woff + cc + i * 16 * BK);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(tA + aoff + cc + i * 16 * BK);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the requests ahead of the products
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], af[ks][j], acc[i][j], 0, 0, 0);
  };

  // multiply state.  (Loading an epilogue's side tensor -- EPI_BN_TAIL / BN_PRE / RESIDUAL / GELU_BWD -- ahead of the
  // tile's last k-tile, to hide its latency behind the products, measured SLOWER: 32 more live registers push the
  // 128-wide kernels to the 256-register limit and the 64-wide ones out of their third block per CU.)
  int c_tile = vblock, c_kt = 0;
  auto finish_ktile = [&](bf16* tA, bf16* tW) {
    if (++c_kt == nk) {  // output tile complete
      const int rt = unit_tile(c_tile);
      const int tile_m = rt / tiles_n, tile_n = rt - tile_m * tiles_n;
      if (KS > 1) {  // this slice's fp32 partial; bias and activation belong to the pass that adds the slices
        float* part = reinterpret_cast<float*>(out) + (size_t)(c_tile - rt * KS) * g.M * g.ldc;
        nt_epilogue<MODE, WI>(g, acc, tile_m * BM_, tile_n * TN_, wm, wn, lane, nullptr, part, 0, 1, 0);
      } else if (out_f32) {
        nt_epilogue<MODE, WI>(g, acc, tile_m * BM_, tile_n * TN_, wm, wn, lane, bias, out, relu, out_f32, beta);
      } else {
        // the k-tile that was just multiplied is free once every wave has finished reading it (LDS reads only:
        // the stages in flight keep streaming while the tile is written out); a wave's private image of its part
        // of the output goes into the source k-tile, or into the weight k-tile when the source one is full
        lds_barrier();
        constexpr int PW = 64 * WI * 16;
        bf16* img = wave * PW < ASZ ? tA + wave * PW : tW + (wave * PW - ASZ);
        // two copies of the epilogue: a bias load (to registers) makes the compiler wait for every load in flight,
        // the staged k-tiles included, and a shared copy would wait on the bias-free path too
        if (bias || (EPI != EPI_NONE && EPI != EPI_BN_TAIL))  // (the other epilogues are register-bound: one copy)
          nt_epilogue_lds<MODE, WI, EPI, true>(g, acc, tile_m * BM_, tile_n * TN_, wm, wn, lane, bias,
                                               reinterpret_cast<bf16*>(out), relu, beta, img, bn_partial, tail_y,
                                               tail_bits);
        else
          nt_epilogue_lds<MODE, WI, EPI, false>(g, acc, tile_m * BM_, tile_n * TN_, wm, wn, lane, nullptr,
                                                reinterpret_cast<bf16*>(out), relu, beta, img, bn_partial, tail_y,
                                                tail_bits);
      }
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      c_kt = 0;
      c_tile += G;
    }
  };

  if constexpr (NA == 2) {
    // one k-tile of lead for both operands; every barrier waits for everything outstanding
    if (vblock < ntiles) {
      VC_STAGE_A(sA0);
      VC_STAGE_W(sW0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int ib = 0;
    while (c_tile < ntiles) {
      if (ca.tile < ntiles) {
        VC_STAGE_A(sA0 + (ib ^ 1) * ASZ);
        VC_STAGE_W(sW0 + (ib ^ 1) * WSZ);
      }
      compute(sA0 + ib * ASZ, sW0 + ib * WSZ);
      finish_ktile(sA0 + ib * ASZ, sW0 + ib * WSZ);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      ib ^= 1;
    }
  } else {
    // Iteration k multiplies k-tile k while the weights of k + 1 and the source rows of k + 2 stream in.  The source
    // rows are requested LAST, so "at most 4 loads outstanding" (a thread stages 4 source rows per k-tile) means that
    // everything for k + 1 has landed; the barrier then makes every wave's share of it visible.  Buffers are reused
    // one iteration after their last read, behind that barrier.  Stores and loads of the epilogue may also be
    // outstanding at the wait: loads complete in order among themselves, so the bound still covers the staged ones.
    bool an = false;
    if (ca.tile < ntiles) VC_STAGE_A(sA0);
    if (cw.tile < ntiles) VC_STAGE_W(sW0);
    if (ca.tile < ntiles) {
      VC_STAGE_A(sA0 + ASZ);
      an = true;
    }
    if (an) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    int ia = 0, iw = 0;
    while (c_tile < ntiles) {
      if (cw.tile < ntiles) VC_STAGE_W(sW0 + (iw ^ 1) * WSZ);
      an = ca.tile < ntiles;
      if (an) VC_STAGE_A(sA0 + (ia == 0 ? 2 : ia - 1) * ASZ);
      compute(sA0 + ia * ASZ, sW0 + iw * WSZ);
      finish_ktile(sA0 + ia * ASZ, sW0 + iw * WSZ);
      if (an) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      ia = ia == 2 ? 0 : ia + 1;
      iw ^= 1;
    }
  }
#undef VC_STAGE_A
#undef VC_STAGE_W
}

// ----------------------------------------------------------------------------
// TN kernel (weight gradient).  g is the FORWARD geometry (src = x).
// ----------------------------------------------------------------------------
__device__ __forceinline__ int tn_swz(int row) { return (row & 3) | ((row >> 1) & 4); }
// element offset of the 8-byte piece p (0..3) of 16-column block `blk` (0..7) of row `row`
__device__ __forceinline__ int tn_tr_off(int row, int blk, int p) {
  return row * 128 + ((blk ^ tn_swz(row)) << 4) + p * 4;
}

__device__ __forceinline__ bf16x8 tr_read_frag(const bf16* tile, int krow0, int blk, int lane) {
  // 16-lane group gq reads rows krow0 + 8*gq + {0..3} then + {4..7}; lane 4q+p of the
  // group supplies row q, columns 4p..4p+3; lane i receives column i of those rows.
  const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int r0 = krow0 + 8 * gq + q;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + tn_tr_off(r0, blk, p)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + tn_tr_off(r0 + 4, blk, p)));
  union {
    struct {
      s16x4 a, b;
    } s;
    bf16x8 v;
  } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

template <bool GLDS>
__global__ __launch_bounds__(NTHR, 2) void igemm_tn_wgrad_bf16_kernel(ConvGeom g, const bf16* __restrict__ x,
                                                                      const bf16* __restrict__ dy, int ldy,
                                                                      float* __restrict__ dw, int m_per_split,
                                                                      int ntiles, int tiles_col, size_t split_stride) {
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * 2 * BK * 128];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wm = wave & 1;  // wn: co half, wm: col half
  int logical;
  xcd_remap(logical);
  const int split = logical / ntiles, tile = logical - split * ntiles;
  dw += (size_t)split * split_stride;  // deterministic mode: a copy of the gradient per split, added in order afterwards
  const int tile_co = tile / tiles_col, tile_col = tile - tile_co * tiles_col;
  const int co0 = tile_co * 128, col0 = tile_col * 128;
  const int m_begin = split * m_per_split;
  const int m_end = min(g.M, m_begin + m_per_split);

  // staging role: LDS row lrow (+16*i), physical 16-byte slot pchunk, which holds logical column chunk
  // lchunk (32-byte blocks are XOR-swizzled by row; 16*i leaves the row's swizzle unchanged)
  const int pchunk = tid & 15, lrow = tid >> 4;
  const int lchunk = ((((pchunk >> 1) ^ tn_swz(lrow)) << 1) | (pchunk & 1));
  // this thread's gather column is fixed for the whole kernel
  const int col = col0 + lchunk * 8;
  const bool colvalid = col < g.K;
  int rd, rh, rw, c;
  decode_chunk(g, col >> 3, rd, rh, rw, c);
  const int co_ld = co0 + lchunk * 8;
  const bool covalid = co_ld < g.Co;

  bf16x8 xreg[4], yreg[4];
  auto stage_dma = [&](int mt, int buf) {
    bf16* sY = smem + buf * (2 * BK * 128);
    bf16* sX = sY + BK * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mt + lrow + 16 * i;
      const void* ysrc = vc_zero_src;
      const void* xsrc = vc_zero_src;
      if (m < m_end) {
        if (covalid) ysrc = dy + (size_t)m * ldy + co_ld;
        if (colvalid) {
          const RowPos rp = decode_row<0>(g, m);
          const int pix = gather_pixel<0>(g, rp, rd, rh, rw);
          if (pix >= 0) xsrc = x + (size_t)pix * g.ldi + c;
        }
      }
      // this wave's instruction fills rows [16*i + 4*wave, +4): 4 rows x 256 B
      glds16(ysrc, sY + (16 * i + 4 * wave) * 128);
      glds16(xsrc, sX + (16 * i + 4 * wave) * 128);
    }
  };
  auto load_tile = [&](int mt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mt + lrow + 16 * i;
      bf16x8 xv = {0, 0, 0, 0, 0, 0, 0, 0}, yv = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < m_end) {
        if (covalid) yv = *reinterpret_cast<const bf16x8*>(dy + (size_t)m * ldy + co_ld);
        if (colvalid) {
          const RowPos rp = decode_row<0>(g, m);
          const int pix = gather_pixel<0>(g, rp, rd, rh, rw);
          if (pix >= 0) xv = *reinterpret_cast<const bf16x8*>(x + (size_t)pix * g.ldi + c);
        }
      }
      xreg[i] = xv;
      yreg[i] = yv;
    }
  };
  auto store_tile = [&](int buf) {
    bf16* sY = smem + buf * (2 * BK * 128);
    bf16* sX = sY + BK * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lrow + 16 * i;
      *reinterpret_cast<bf16x8*>(sY + row * 128 + pchunk * 8) = yreg[i];
      *reinterpret_cast<bf16x8*>(sX + row * 128 + pchunk * 8) = xreg[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (m_begin < m_end) {
    if (GLDS) {
      stage_dma(m_begin, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      load_tile(m_begin);
      store_tile(0);
    }
  }
  __syncthreads();
  int buf = 0;
  for (int mt = m_begin; mt < m_end; mt += BK) {
    const bool more = mt + BK < m_end;
    if (more) {
      if (GLDS) stage_dma(mt + BK, buf ^ 1);
      else load_tile(mt + BK);
    }
    const bf16* sY = smem + buf * (2 * BK * 128);
    const bf16* sX = sY + BK * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 yf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        yf[i] = tr_read_frag(sY, ks * 32, wn * 4 + i, lane);
        xf[i] = tr_read_frag(sX, ks * 32, wm * 4 + i, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (GLDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (more) {
      store_tile(buf ^ 1);
    }
    __syncthreads();
    buf ^= 1;
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wn * 64 + i * 16 + (lane >> 4) * 4 + r;
      if (co >= g.Co) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cc = col0 + wm * 64 + j * 16 + (lane & 15);
        if (cc < g.K) atomicAdd(dw + (size_t)co * g.K + cc, acc[i][j][r]);
      }
    }
  }
}

// Weight-gradient kernel with the per-row index arithmetic taken out of the staging loop: the 64 pixel rows
// of a k-tile are decoded ONCE (by 64 threads, one tile ahead, into a small LDS table) instead of by every
// thread that stages a piece of them, dy pointers advance by a constant per tile, and the transposing
// fragment reads use per-lane offsets computed before the loop plus immediates.
struct RowInfo {
  int off;         // element offset of the row's window origin (tap 0,0,0) in x; may be negative
  int d0, h0, w0;  // window origin coordinates; h0 = -2^28 marks a row beyond the split
};

// Block swizzle of a pixel-major LDS tile whose rows are W elements (W/16 blocks of 16 columns): the value is
// XORed into the block index.  It is chosen so that the 32 lanes one transposing read serves (rows 8*gq + q,
// gq in {0,1}, q in 0..3, four 8-byte pieces each) fall on 64 different banks.
template <int W>
__device__ __forceinline__ int tile_swz(int row) {
  if (W == 128) return (row & 3) | ((row >> 1) & 4);
  return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);  // W == 64: a row is half a bank sweep
}

// TCO x TCOL output tile (output channels x gather columns), each 128 or 64: the 64-wide forms serve layers
// with <= 64 output channels or <= 64 reduction columns, which would leave half of a 128-wide tile empty.
template <int TCO, int TCOL>
__global__ __launch_bounds__(NTHR, 2) void igemm_tn_wgrad_fast_kernel(ConvGeom g, const bf16* __restrict__ x,
                                                                      const bf16* __restrict__ dy, int ldy,
                                                                      float* __restrict__ dw, int m_per_split,
                                                                      int ntiles, int tiles_col,
                                                                      float* __restrict__ dbias, size_t split_stride) {
  constexpr int STAGE = BK * (TCO + TCOL);  // elements per pipeline stage: dY tile then X tile
  constexpr int LY = TCO / 8, RY = NTHR / LY, PY = BK / RY;     // dY: lanes per row, rows per pass, passes
  constexpr int LX = TCOL / 8, RX = NTHR / LX, PX = BK / RX;    // X likewise
  constexpr int WI = TCO / 32, WJ = TCOL / 32;                  // 16x16 tiles per wave
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * STAGE];
  __shared__ __attribute__((aligned(16))) RowInfo rinfo[2][BK];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wm = wave & 1;  // wn: co half, wm: col half
  int logical;
  xcd_remap(logical);
  const int split = logical / ntiles, tile = logical - split * ntiles;
  if (split_stride) {  // deterministic mode: a copy of both gradients per split, added in order afterwards
    dw += (size_t)split * split_stride;
    if (dbias) dbias += (size_t)split * g.Co;
  }
  const int tile_co = tile / tiles_col, tile_col = tile - tile_co * tiles_col;
  const int co0 = tile_co * TCO, col0 = tile_col * TCOL;
  const int m_begin = split * m_per_split;
  const int m_end = min(g.M, m_begin + m_per_split);
  const bf16* zero = reinterpret_cast<const bf16*>(vc_zero_src);

  // staging roles (the LDS image is lane-linear per wave instruction; the swizzle is applied to the source)
  const int ypc = tid % LY, yrow = tid / LY;
  const int ychunk = ((((ypc >> 1) ^ tile_swz<TCO>(yrow)) << 1) | (ypc & 1));
  const int xpc = tid % LX, xrow = tid / LX;
  const int xchunk = ((((xpc >> 1) ^ tile_swz<TCOL>(xrow)) << 1) | (xpc & 1));
  // this thread's gather column (tap, channel chunk) is fixed for the whole kernel
  const int col = col0 + xchunk * 8;
  const bool colvalid = col < g.K;
  int rd, rh, rw, c;
  decode_chunk(g, col >> 3, rd, rh, rw, c);
  const int tapoff = ((rd * g.Hi + rh) * g.Wi + rw) * g.ldi + c;
  const int co_ld = co0 + ychunk * 8;
  const bool covalid = co_ld < g.Co;

  auto decode_rows = [&](int mt, int slot, int t) {  // t in [0, BK): one row per calling thread
    const int m = mt + t;
    RowInfo ri;
    ri.off = 0;
    ri.d0 = ri.w0 = 0;
    ri.h0 = -(1 << 28);
    if (m < m_end) {
      const RowPos rp = decode_row<0>(g, m);
      ri.d0 = rp.d0;
      ri.h0 = rp.h0;
      ri.w0 = rp.w0;
      ri.off = (((rp.n * g.Di + rp.d0) * g.Hi + rp.h0) * g.Wi + rp.w0) * g.ldi;
    }
    rinfo[slot][t] = ri;
  };

  const bf16* yp[PY];
#pragma unroll
  for (int i = 0; i < PY; ++i) yp[i] = dy + (size_t)(m_begin + yrow + RY * i) * ldy + co_ld;
  const int yinc = BK * ldy;

  auto stage = [&](int mt, int buf) {
    bf16* sY = smem + buf * STAGE;
    bf16* sX = sY + BK * TCO;
    int4 ri[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) ri[i] = *reinterpret_cast<const int4*>(&rinfo[buf][xrow + RX * i]);
#pragma unroll
    for (int i = 0; i < PY; ++i) {
      const int mvalid = (mt + yrow + RY * i < m_end) & (int)covalid;
      const bf16* ysrc = mvalid ? yp[i] : zero;
      glds16(ysrc, sY + (RY * i + (RY / 4) * wave) * TCO);
      yp[i] += yinc;
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      // branch-free: every test is evaluated, the pointer is selected
      const int xok = (int)colvalid & (int)((unsigned)(ri[i].y + rd) < (unsigned)g.Di) &
                      (int)((unsigned)(ri[i].z + rh) < (unsigned)g.Hi) &
                      (int)((unsigned)(ri[i].w + rw) < (unsigned)g.Wi);
      const int xo = xok ? ri[i].x + tapoff : 0;
      const bf16* xsrc = xok ? x + xo : zero;
      glds16(xsrc, sX + (RX * i + (RX / 4) * wave) * TCOL);
    }
  };

  f32x4 acc[WI][WJ];
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposing fragment reads: lane (gq, q, p) reads rows 8*gq + q (+4) (+32*ks) of a 16-column block; the
  // rows' block swizzle is the same for all of them
  const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int r0 = 8 * gq + q;
  int yo[WI], xo[WJ];
#pragma unroll
  for (int i = 0; i < WI; ++i) yo[i] = r0 * TCO + p * 4 + (((wn * WI + i) ^ tile_swz<TCO>(r0)) << 4);
#pragma unroll
  for (int j = 0; j < WJ; ++j) xo[j] = BK * TCO + r0 * TCOL + p * 4 + (((wm * WJ + j) ^ tile_swz<TCOL>(r0)) << 4);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  auto frag = [&](const bf16* tile, int off, int w) {
    union {
      struct {
        s16x4 a, b;
      } s;
      bf16x8 v;
    } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off));
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off + 4 * w));
    return u.v;
  };
  // Bias gradient = column sums of dy = dy^T x (a column of ones): the waves that own column tile 0 multiply
  // their dy fragments by an all-ones operand as well (one extra MFMA per fragment, no extra loads).
  const bool do_bias = dbias != nullptr && tile_col == 0 && wm == 0;
  f32x4 bacc[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};
  auto compute = [&](const bf16* tile) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 yf[WI], xf[WJ];
#pragma unroll
      for (int i = 0; i < WI; ++i) yf[i] = frag(tile, yo[i] + ks * 32 * TCO, TCO);
#pragma unroll
      for (int j = 0; j < WJ; ++j) xf[j] = frag(tile, xo[j] + ks * 32 * TCOL, TCOL);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < WI; ++i) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], ones, bacc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  };

  // row tables of the first two tiles, then tile 0
  if (tid < 2 * BK) decode_rows(m_begin + (tid >> 6) * BK, tid >> 6, tid & 63);
  __syncthreads();
  if (m_begin < m_end) stage(m_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int mt = m_begin; mt < m_end; mt += 2 * BK) {
    // tile mt in buffer 0
    if (mt + BK < m_end) stage(mt + BK, 1);
    if (tid < BK) decode_rows(mt + 2 * BK, 0, tid);  // table slot 0 was last read when tile mt was staged
    compute(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (mt + BK >= m_end) break;
    // tile mt + BK in buffer 1
    if (mt + 2 * BK < m_end) stage(mt + 2 * BK, 0);
    if (tid < BK) decode_rows(mt + 3 * BK, 1, tid);
    compute(smem + STAGE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < WI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wn * (TCO / 2) + i * 16 + (lane >> 4) * 4 + r;
      if (co >= g.Co) continue;
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        const int cc = col0 + wm * (TCOL / 2) + j * 16 + (lane & 15);
        if (cc < g.K) atomicAdd(dw + (size_t)co * g.K + cc, acc[i][j][r]);
      }
    }
  }
  if (do_bias && (lane & 15) == 0) {  // every column of bacc holds the same sums: lane column 0 adds them
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wn * (TCO / 2) + i * 16 + (lane >> 4) * 4 + r;
        if (co < g.Co) atomicAdd(dbias + co, bacc[i][r]);
      }
  }
}

// ----------------------------------------------------------------------------
// Naive kernels (fp32 path and cross-check).
// ----------------------------------------------------------------------------
template <typename T, int MODE>
__global__ void conv_naive_kernel(ConvGeom g, const T* __restrict__ src, const T* __restrict__ wgt,
                                  const float* __restrict__ bias, void* __restrict__ out, int relu, int out_f32,
                                  int beta) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)g.M * g.Co) return;
  const int m = (int)(idx / g.Co), co = (int)(idx - (int64_t)m * g.Co);
  const RowPos rp = decode_row<MODE>(g, m);
  float acc = 0.f;
  const T* wrow = wgt + (size_t)co * g.K;
  int tap = 0;
  for (int rd = 0; rd < g.kd; ++rd)
    for (int rh = 0; rh < g.kh; ++rh)
      for (int rw = 0; rw < g.kw; ++rw, ++tap) {
        const int pix = gather_pixel<MODE>(g, rp, rd, rh, rw);
        if (pix < 0) continue;
        const T* sp = src + (size_t)pix * g.ldi;
        const T* wp = wrow + (size_t)tap * g.Ci;
        for (int c = 0; c < g.Ci; ++c) acc = fmaf(to_f32(sp[c]), to_f32(wp[c]), acc);
      }
  if (bias) acc += bias[co];
  const size_t o = (size_t)m * g.ldc + co;
  if (out_f32) {
    float* op = reinterpret_cast<float*>(out);
    if (beta) acc += op[o];
    if (relu) acc = fmaxf(acc, 0.f);
    op[o] = acc;
  } else {
    T* op = reinterpret_cast<T*>(out);
    if (beta) acc += to_f32(op[o]);
    if (relu) acc = fmaxf(acc, 0.f);
    op[o] = from_f32<T>(acc);
  }
}

template <typename T>
__global__ void wgrad_naive_kernel(ConvGeom g, const T* __restrict__ x, const T* __restrict__ dy, int ldy,
                                   float* __restrict__ dw) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)g.Co * g.K) return;
  const int co = (int)(idx / g.K), kk = (int)(idx - (int64_t)co * g.K);
  const int tap = kk / g.Ci, c = kk - tap * g.Ci;
  const int rw = tap % g.kw, rh = (tap / g.kw) % g.kh, rd = tap / (g.kw * g.kh);
  float acc = 0.f;
  for (int m = 0; m < g.M; ++m) {
    const RowPos rp = decode_row<0>(g, m);
    const int pix = gather_pixel<0>(g, rp, rd, rh, rw);
    if (pix < 0) continue;
    acc = fmaf(to_f32(dy[(size_t)m * ldy + co]), to_f32(x[(size_t)pix * g.ldi + c]), acc);
  }
  dw[idx] += acc;
}

// ----------------------------------------------------------------------------
// Weight packing: torch layout [Co][Ci][taps] fp32 <-> packed [Co][taps][Cip] (forward)
// and [Ci][taps][Cop] (data gradient), in the compute type.
// ----------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wt, int Co, int Ci,
                                   int taps, int Cip, int Cop) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // forward pack: [Co][taps][Cip]
  const int64_t nf = (int64_t)Co * taps * Cip;
  if (wf && idx < nf) {
    const int c = (int)(idx % Cip);
    const int tap = (int)((idx / Cip) % taps);
    const int co = (int)(idx / ((int64_t)Cip * taps));
    wf[idx] = from_f32<T>(c < Ci ? w[((size_t)co * Ci + c) * taps + tap] : 0.f);
  }
  // transposed pack: [Ci][taps][Cop]
  const int64_t nt = (int64_t)Ci * taps * Cop;
  if (wt && idx < nt) {
    const int co = (int)(idx % Cop);
    const int tap = (int)((idx / Cop) % taps);
    const int ci = (int)(idx / ((int64_t)Cop * taps));
    wt[idx] = from_f32<T>(co < Co ? w[((size_t)co * Ci + ci) * taps + tap] : 0.f);
  }
}

// data-gradient pack of a convolution whose kernel covers its whole input (a Linear layer over the
// flattened channels-last map): wt[(tap*Cip + c)][Cop]
template <typename T>
__global__ void pack_weight_flat_t_kernel(const float* __restrict__ w, T* __restrict__ wt, int Co, int Ci, int taps,
                                          int Cip, int Cop) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)taps * Cip * Cop) return;
  const int co = (int)(idx % Cop);
  const int c = (int)((idx / Cop) % Cip);
  const int tap = (int)(idx / ((int64_t)Cop * Cip));
  wt[idx] = from_f32<T>((co < Co && c < Ci) ? w[((size_t)co * Ci + c) * taps + tap] : 0.f);
}

// Width-unfolded first layer: torch weight [Co][Ci][to][kw] (to = kd*kh outer taps)  ->  wf[Cop][to][Jp] with
// channel index q = j*Ci + c (zero for q >= kw*Ci and for padded output channels)
template <typename T>
__global__ void pack_weight_unfold_kernel(const float* __restrict__ w, T* __restrict__ wf, int Co, int Ci, int to,
                                          int kw, int Jp, int Cop) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Cop * to * Jp) return;
  const int q = (int)(idx % Jp);
  const int t = (int)((idx / Jp) % to);
  const int co = (int)(idx / ((int64_t)Jp * to));
  float v = 0.f;
  if (co < Co && q < kw * Ci) {
    const int j = q / Ci, c = q - j * Ci;
    v = w[(((size_t)co * Ci + c) * to + t) * kw + j];
  }
  wf[idx] = from_f32<T>(v);
}

// the inverse for the gradient: dw[co][c][t][j] (+)= dwp[co][t][j*Ci + c]; flags as for unpack_wgrad_kernel
__global__ void unpack_wgrad_unfold_kernel(float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int to,
                                           int kw, int Jp, int flags) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Co * Ci * to * kw) return;
  const int j = (int)(idx % kw);
  const int t = (int)((idx / kw) % to);
  const int c = (int)((idx / ((int64_t)kw * to)) % Ci);
  const int co = (int)(idx / ((int64_t)kw * to * Ci));
  const size_t src = ((size_t)co * to + t) * Jp + j * Ci + c;
  const float v = dwp[src];
  if (flags & 2) dwp[src] = 0.f;
  dw[idx] = (flags & 1) ? dw[idx] + v : v;
}

// packed fp32 gradient [Co][taps][Cip] -> torch layout [Co][Ci][taps] (accumulating or not)
// flags: bit 0 = add to dw instead of overwriting it, bit 1 = zero the packed source after reading it (a
// caller-owned accumulator is then ready for the next weight-gradient launch without a fill pass)
__global__ void unpack_wgrad_kernel(float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int taps,
                                    int Cip, int flags) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Co * Ci * taps) return;
  const int tap = (int)(idx % taps);
  const int ci = (int)((idx / taps) % Ci);
  const int co = (int)(idx / ((int64_t)taps * Ci));
  const size_t src = ((size_t)co * taps + tap) * Cip + ci;
  const float v = dwp[src];
  if (flags & 2) dwp[src] = 0.f;
  dw[idx] = (flags & 1) ? dw[idx] + v : v;
}

// ---- the same three re-layouts for layers with many taps (a Linear over a flattened feature map: 144 taps x 256
// channels x 1024 outputs = 151 MB of fp32): 32 x 32 tiles through LDS, so that reads and writes both run along the
// contiguous index of their side.  The element-wise forms above gather with a stride of `taps` floats. ----
constexpr int TT = 32;
// wf[co][t][c] = w[co][c][t] (zero for c >= Ci); grid (taps/32, Cip/32, Co)
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_tiled_kernel(const float* __restrict__ w, T* __restrict__ wf, int Ci,
                                                                 int taps, int Cip) {
  __shared__ float tile[TT][TT + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int co = blockIdx.z, c0 = blockIdx.y * TT, t0 = blockIdx.x * TT;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    tile[ty + 8 * i][tx] = (c < Ci && t < taps) ? w[((size_t)co * Ci + c) * taps + t] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    if (t < taps && c < Cip) wf[((size_t)co * taps + t) * Cip + c] = from_f32<T>(tile[tx][ty + 8 * i]);
  }
}

// wt[(t * Cip + c)][co] = w[co][c][t] (zero for co >= Co, c >= Ci); grid (taps/32, Cop/32, Cip)
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_flat_t_tiled_kernel(const float* __restrict__ w, T* __restrict__ wt,
                                                                        int Co, int Ci, int taps, int Cip, int Cop) {
  __shared__ float tile[TT][TT + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.z, co0 = blockIdx.y * TT, t0 = blockIdx.x * TT;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + ty + 8 * i, t = t0 + tx;
    tile[ty + 8 * i][tx] = (co < Co && c < Ci && t < taps) ? w[((size_t)co * Ci + c) * taps + t] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, co = co0 + tx;
    if (t < taps && co < Cop) wt[((size_t)t * Cip + c) * Cop + co] = from_f32<T>(tile[tx][ty + 8 * i]);
  }
}

// dw[co][c][t] (+)= dwp[co][t][c]; flags as for unpack_wgrad_kernel; grid (taps/32, Ci/32, Co)
__global__ __launch_bounds__(256) void unpack_wgrad_tiled_kernel(float* __restrict__ dwp, float* __restrict__ dw, int Ci,
                                                                  int taps, int Cip, int flags) {
  __shared__ float tile[TT][TT + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int co = blockIdx.z, c0 = blockIdx.y * TT, t0 = blockIdx.x * TT;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    float v = 0.f;
    if (t < taps && c < Ci) {
      const size_t src = ((size_t)co * taps + t) * Cip + c;
      v = dwp[src];
      if (flags & 2) dwp[src] = 0.f;
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    if (c < Ci && t < taps) {
      const size_t dst = ((size_t)co * Ci + c) * taps + t;
      const float v = tile[tx][ty + 8 * i];
      dw[dst] = (flags & 1) ? dw[dst] + v : v;
    }
  }
}

// y[m][c] = act(sum_s partial[s][m][c] + bias[c]): the pass that adds the slices of a split reduction
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                     T* __restrict__ y, int64_t n4, int C4, int S, int64_t slice, int relu) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 a = *reinterpret_cast<const f32x4*>(partial + i * 4);
    for (int s2 = 1; s2 < S; ++s2) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(partial + s2 * slice + i * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] += b[r];
    }
    if (bias) {
      const int c = (int)(i % C4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] += bias[c + r];
    }
    if (relu) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.f);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) y[i * 4 + r] = from_f32<T>(a[r]);
  }
}

// the tiled forms pay once a (channel, tap) plane is a few tiles large
bool relayout_tiled(int co, int ci, int taps) { return taps >= 16 && ci >= 16 && co <= 65535 && ci <= 65535; }

int fill_geom(ConvGeom& g, const int* d, int mode, const char* who) {
  // d: n, di, hi, wi, ci, do, ho, wo, co, kd, kh, kw, sd, sh, sw, pd, ph, pw  (forward meaning)
  const int n = d[0], di = d[1], hi = d[2], wi = d[3], ci = d[4], dout = d[5], ho = d[6], wo = d[7], co = d[8];
  const int kd = d[9], kh = d[10], kw = d[11], sd = d[12], sh = d[13], sw = d[14], pd = d[15], ph = d[16], pw = d[17];
  VC_CHECK_SHAPE(n > 0 && di > 0 && hi > 0 && wi > 0 && ci > 0 && dout > 0 && ho > 0 && wo > 0 && co > 0,
                 "%s: non-positive extent", who);
  VC_CHECK_SHAPE(kd > 0 && kh > 0 && kw > 0 && sd > 0 && sh > 0 && sw > 0 && pd >= 0 && ph >= 0 && pw >= 0,
                 "%s: bad kernel/stride/pad", who);
  VC_CHECK_SHAPE((di + 2 * pd - kd) / sd + 1 == dout && (hi + 2 * ph - kh) / sh + 1 == ho &&
                     (wi + 2 * pw - kw) / sw + 1 == wo,
                 "%s: output extent does not match input/kernel/stride/pad", who);
  VC_CHECK_SHAPE((int64_t)n * di * hi * wi * ci < (1ll << 31) && (int64_t)n * dout * ho * wo * co < (1ll << 31),
                 "%s: tensor exceeds 2^31 elements", who);
  g.ksplit = 1;
  g.kd = kd; g.kh = kh; g.kw = kw;
  g.sd = sd; g.sh = sh; g.sw = sw;
  g.pd = pd; g.ph = ph; g.pw = pw;
  g.N = n;
  if (mode == 0) {  // forward (also the geometry the weight-gradient kernels use)
    g.Di = di; g.Hi = hi; g.Wi = wi; g.Ci = ci;
    g.Do = dout; g.Ho = ho; g.Wo = wo;
    g.Co = co;
  } else {  // data gradient: rows are dx pixels, source is dy
    VC_CHECK_SHAPE((sd & (sd - 1)) == 0 && (sh & (sh - 1)) == 0 && (sw & (sw - 1)) == 0,
                   "%s: data gradient needs power-of-two strides", who);
    g.Di = dout; g.Hi = ho; g.Wi = wo; g.Ci = co;
    g.Do = di; g.Ho = hi; g.Wo = wi;
    g.Co = ci;
  }
  g.K = kd * kh * kw * g.Ci;
  g.ldw = g.K;
  g.M = g.N * g.Do * g.Ho * g.Wo;
  g.ldc = g.Co;
  g.ldi = g.Ci;
  g.fWo = make_fastdiv(g.Wo);
  g.fHo = make_fastdiv(g.Ho);
  g.fDo = make_fastdiv(g.Do);
  g.fC8 = make_fastdiv(g.Ci / 8 > 0 ? g.Ci / 8 : 1);
  g.fkw = make_fastdiv(kw);
  g.fkh = make_fastdiv(kh);
  return VC_OK;
}

// VC_IGEMM_STAGE=reg selects the register-staged variants (A/B measurements); default is LDS-DMA.
bool use_glds() {
  static const int v = [] {
    const char* e = getenv("VC_IGEMM_STAGE");
    return (e && strcmp(e, "reg") == 0) ? 0 : 1;
  }();
  return v != 0;
}

// VC_IGEMM_NT = generic | fast | persist selects the NT kernel family (A/B measurements); default persist
int nt_variant() {
  static const int v = [] {
    const char* e = getenv("VC_IGEMM_NT");
    if (e && strcmp(e, "generic") == 0) return 0;
    if (e && strcmp(e, "fast") == 0) return 1;
    return 2;
  }();
  return v;
}
// widest channel count that still takes the 64-channel tile (VC_NT_NARROW overrides, for tuning runs)
int nt_narrow() {
  static const int v = [] {
    const char* e = getenv("VC_NT_NARROW");
    return e ? atoi(e) : 64;
  }();
  return v;
}
int nt_persist_blocks() {
  static const int v = [] {
    const char* e = getenv("VC_NT_BLOCKS");
    const int n = e ? atoi(e) : 0;
    return n > 0 ? n : 512;
  }();
  return v;
}

// VC_IGEMM_NT=generic keeps every layer on the general kernel (A/B measurements)
bool use_generic_nt() {
  static const int v = [] {
    const char* e = getenv("VC_IGEMM_NT");
    return (e && strcmp(e, "generic") == 0) ? 1 : 0;
  }();
  return v != 0;
}

int wgrad_narrow() {
  static const int v = [] {
    const char* e = getenv("VC_WGRAD_NARROW");
    return e ? atoi(e) : 1;
  }();
  return v;
}

// number of blocks the weight-gradient launch aims for (VC_WGRAD_BLOCKS overrides, for tuning runs)
int wgrad_target_blocks() {
  static const int v = [] {
    const char* e = getenv("VC_WGRAD_BLOCKS");
    const int n = e ? atoi(e) : 0;
    return n > 0 ? n : 1024;
  }();
  return v;
}

// fewest pixel rows one split of the weight gradient should cover (VC_WGRAD_SPLIT_ROWS overrides, for tuning runs)
int wgrad_min_rows_per_split() {
  static const int v = [] {
    const char* e = getenv("VC_WGRAD_SPLIT_ROWS");
    const int n = e ? atoi(e) : 0;
    return n > 0 ? n : 2048;
  }();
  return v;
}

bool nt_persist_eligible(const ConvGeom& g, int dtype) {
  const bool single_tap = g.kd * g.kh * g.kw == 1;
  return dtype == VC_BF16 && use_glds() && g.ldi % 8 == 0 && g.K >= 8 && nt_variant() == 2 &&
         (g.Ci % BK == 0 || (single_tap && g.Ci % 8 == 0));
}

// Slices a single-tap layer's reduction should be cut into so that a launch fills the chip (1 = leave it whole):
// few output tiles, a long reduction (CRNN's fc1: 4 x 8 tiles of 128 x 128, 36864 channels -> 16 slices of 36 k-tiles)
int ksplit_plan(const ConvGeom& g, int dtype) {
  if (!nt_persist_eligible(g, dtype) || g.kd * g.kh * g.kw != 1 || g.Ci % BK || g.Co % 4 || g.ldc != g.Co ||
      g.Co <= nt_narrow())
    return 1;
  const int tiles = ceil_div(g.M, BM) * ceil_div(g.Co, BN);
  const int ncb = g.Ci / BK;
  int S = 1;
  // two work units per CU as for whole layers; more slices only add partial-sum traffic
  while (S < 32 && tiles * S < 512 && ncb % (S * 2) == 0 && ncb / (S * 2) >= 16) S *= 2;
  return tiles <= 128 ? S : 1;
}

// VC_IGEMM_BL=0 keeps the global-load staging of the persistent kernel everywhere, 2 uses the buffer form for
// single-tap layers too (A/B measurements); default 1 = buffer form for multi-tap convolutions
int use_bl() {
  static const int v = [] {
    const char* e = getenv("VC_IGEMM_BL");
    return e ? atoi(e) : 1;
  }();
  return v;
}

// VC_NT_RING = 2 | 3: staged source k-tiles of the persistent kernel (A/B measurements); default 3
int nt_ring() {
  static const int v = [] {
    const char* e = getenv("VC_NT_RING");
    return (e && atoi(e) == 2) ? 2 : 3;
  }();
  return v;
}

// Layers of up to 64 output channels run on 256 x 64 tiles (measured on the 56 x 56 ResNet layers: 4-8 % faster than
// 128 x 64, forward and data gradient); VC_NT_TALL=0 keeps the 128 x 64 tile (read at every call: A/B measurements)
int nt_tall() {
  const char* e = getenv("VC_NT_TALL");
  return e ? atoi(e) : 1;
}

// geometries the buffer-load staging covers (see igemm_nt_persist_kernel)
bool bl_eligible(const ConvGeom& g, int mode, bool single_tap_too = false) {
  // Measured on the ResNet-50 layer table: 3x3 layers gain 4-18 % (forward 64 -> 64: 0.267 -> 0.256 ms, its data
  // gradient 0.301 -> 0.253 ms), single-tap layers lose 0-5 % (their tiles have no per-tap work to save), so those keep
  // the global-load staging unless VC_IGEMM_BL=2 forces the buffer form (tests, A/B runs).
  const int taps = g.kd * g.kh * g.kw;
  if (!use_bl() || taps > 32 || (taps == 1 && use_bl() != 2 && !single_tap_too)) return false;
  if (mode == 1 && (g.sd != 1 || g.sh != 1 || g.sw != 1)) return false;
  const int64_t abytes = (int64_t)g.N * g.Di * g.Hi * g.Wi * g.ldi * 2, wbytes = (int64_t)g.Co * g.ldw * 2;
  return abytes < 0xFFFF0000ll && wbytes < 0xFFFF0000ll;
}

// launches igemm_nt_persist_kernel<MODE, TN, EPI> with the staging variant the geometry allows
template <int MODE, int TN, int EPI>
void launch_persist(const ConvGeom& g, int grid, hipStream_t stream, const void* src, const void* wgt, const float* bias,
                    void* out, int relu, int out_f32, int beta, int ntiles, float* bn_partial, const void* tail_y,
                    const uint8_t* tail_bits) {
#define VC_PERSIST_LAUNCH(BL_, NA_)                                                                                 \
  hipLaunchKernelGGL((igemm_nt_persist_kernel<MODE, TN, EPI, BL_, NA_>), dim3(grid), dim3(NTHR), 0, stream, g,       \
                     (const bf16*)src, (const bf16*)wgt, bias, out, relu, out_f32, beta, ntiles, bn_partial,         \
                     (const bf16*)tail_y, tail_bits)
  const bool bl = bl_eligible(g, MODE) && g.ksplit == 1;
  // Three staged source k-tiles pay where the source streams from HBM into wide tiles: measured on the ResNet-50 layer
  // table, 1x1 layers of 256..1024 channels at 28 x 28 and 14 x 14 gain 5-12 % (forward and data gradient); the
  // 64-channel tile loses a third (64 KiB of LDS: two blocks per CU instead of three), multi-tap layers lose 5 % (their
  // source re-reads come out of L2) and the 7 x 7 layers 3 %, so those keep two.
  bool deep = false;
  if constexpr (TN == 128 && EPI == EPI_NONE) deep = nt_ring() == 3 && !bl && g.M >= 50000;
  if constexpr (TN == 128 && EPI == EPI_NONE) {
    if (deep) {
      VC_PERSIST_LAUNCH(false, 3);
      return;
    }
  }
  if (bl) VC_PERSIST_LAUNCH(true, 2);
  else VC_PERSIST_LAUNCH(false, 2);
#undef VC_PERSIST_LAUNCH
}

template <int MODE>
int launch_nt(const ConvGeom& g, int dtype, const void* src, const void* wgt, const float* bias, void* out, int relu,
              int out_f32, int beta, hipStream_t stream, const char* who, float* bn_partial = nullptr) {
  if (bn_partial && !(nt_persist_eligible(g, dtype) && !out_f32 && !beta)) {
    vc_set_error("%s: fused BatchNorm statistics need the bf16 persistent kernel (source channels %% 64 == 0, "
                 "bf16 output, beta = 0); ask vc_conv_fwd_stats_supported() first", who);
    return VC_ERR_SHAPE;
  }
  if (dtype == VC_BF16) {
    VC_CHECK_SHAPE(g.Ci % 8 == 0, "%s: bf16 path needs source channels %% 8 == 0 (got %d)", who, g.Ci);
    const int tiles = ceil_div(g.M, BM) * ceil_div(g.Co, BN);
    if (nt_persist_eligible(g, dtype) && MODE != 2 && vc_nt8_eligible(g, MODE, out_f32, EPI_NONE)) {
      vc_nt8_launch(g, MODE, EPI_NONE, stream, src, wgt, bias, out, relu, beta, bn_partial, nullptr, nullptr);
    } else if (nt_persist_eligible(g, dtype)) {
      if (g.Co <= nt_narrow() && MODE != 2 && nt_tall() && !out_f32 && bl_eligible(g, MODE, true)) {
        // 256 x 64 tiles, the waves stacked along the rows (buffer form only: the global-load form spills here)
        const int t = ceil_div(g.M, 256) * ceil_div(g.Co, 64);
        const int grid = t < nt_persist_blocks() ? t : nt_persist_blocks();
        if constexpr (MODE != 2)
          hipLaunchKernelGGL((igemm_nt_persist_kernel<MODE, 64, EPI_NONE, true, 2, 256>), dim3(grid), dim3(NTHR), 0,
                             stream, g, (const bf16*)src, (const bf16*)wgt, bias, out, relu, out_f32, beta, t,
                             bn_partial, (const bf16*)nullptr, (const uint8_t*)nullptr);
      } else if (g.Co <= nt_narrow()) {  // a 128-wide channel tile would be half padding
        const int t64 = ceil_div(g.M, BM) * ceil_div(g.Co, 64);
        const int grid = t64 < nt_persist_blocks() * 3 / 2 ? t64 : nt_persist_blocks() * 3 / 2;
        launch_persist<MODE, 64, EPI_NONE>(g, grid, stream, src, wgt, bias, out, relu, out_f32, beta, t64, bn_partial,
                                           nullptr, nullptr);
      } else {
        const int grid = tiles < nt_persist_blocks() ? tiles : nt_persist_blocks();
        launch_persist<MODE, 128, EPI_NONE>(g, grid, stream, src, wgt, bias, out, relu, out_f32, beta, tiles, bn_partial,
                                            nullptr, nullptr);
      }
    } else if (use_glds() && g.Ci % BK == 0 && g.ldi % 8 == 0 && nt_variant() >= 1)
      hipLaunchKernelGGL((igemm_nt_fast_kernel<MODE>), dim3(tiles), dim3(NTHR), 0, stream, g, (const bf16*)src,
                         (const bf16*)wgt, bias, out, relu, out_f32, beta);
    else if (use_glds() && g.Co <= nt_narrow())  // a 128-wide channel tile would be at least half padding
      hipLaunchKernelGGL((igemm_nt_bf16_kernel<MODE, true, 64>), dim3(ceil_div(g.M, BM) * ceil_div(g.Co, 64)),
                         dim3(NTHR), 0, stream, g, (const bf16*)src, (const bf16*)wgt, bias, out, relu, out_f32, beta);
    else if (use_glds())
      hipLaunchKernelGGL((igemm_nt_bf16_kernel<MODE, true>), dim3(tiles), dim3(NTHR), 0, stream, g, (const bf16*)src,
                         (const bf16*)wgt, bias, out, relu, out_f32, beta);
    else
      hipLaunchKernelGGL((igemm_nt_bf16_kernel<MODE, false>), dim3(tiles), dim3(NTHR), 0, stream, g,
                         (const bf16*)src, (const bf16*)wgt, bias, out, relu, out_f32, beta);
  } else if (dtype == VC_F32) {
    const int64_t total = (int64_t)g.M * g.Co;
    hipLaunchKernelGGL((conv_naive_kernel<float, MODE>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream,
                       g, (const float*)src, (const float*)wgt, bias, out, relu, 1, beta);
  } else {
    vc_set_error("%s: unsupported dtype %d", who, dtype);
    return VC_ERR_DTYPE;
  }
  VC_LAUNCH_CHECK(who);
  return VC_OK;
}

// stride-1 data gradient whose epilogue also finishes the gradient of a residual block's output (see TAIL)
bool dgrad_tail_eligible(const ConvGeom& g, int dtype) {
  return nt_persist_eligible(g, dtype) && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.Co % 8 == 0 && g.ldc == g.Co &&
         g.Co > nt_narrow();
}

}  // namespace

extern "C" {

int vc_conv_fwd(const int* desc, int dtype, const void* x, int ldx, const void* w_packed, const float* bias, void* y,
                int ldy, int relu, int out_f32, int beta, hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 0, "vc_conv_fwd");
  if (rc) return rc;
  VC_CHECK_ARG(x && w_packed && y, "vc_conv_fwd: null pointer");
  if (ldy > 0) g.ldc = ldy;
  if (ldx > 0) g.ldi = ldx;
  VC_CHECK_SHAPE(g.ldc >= g.Co && g.ldi >= g.Ci, "vc_conv_fwd: leading dimension smaller than channel count");
  VC_CHECK_SHAPE(dtype != VC_BF16 || g.ldi % 8 == 0, "vc_conv_fwd: bf16 path needs ldx %% 8 == 0");
  return launch_nt<0>(g, dtype, x, w_packed, bias, y, relu, out_f32, beta, stream, "vc_conv_fwd");
}

// Forward layer with its reduction cut into slices (vc_conv_fwd_splitk_plan(desc, dtype) > 1 of them; 1 = use
// vc_conv_fwd): one launch computes every (output tile, slice) pair into workspace [slices][M][co] fp32, a second
// adds the slices, the bias and the ReLU into y (bf16, or fp32 with out_f32).
int vc_conv_fwd_splitk_plan(const int* desc, int dtype) {
  ConvGeom g;
  if (fill_geom(g, desc, 0, "vc_conv_fwd_splitk_plan")) return 1;
  return ksplit_plan(g, dtype);
}

int vc_conv_fwd_splitk(const int* desc, int dtype, const void* x, const void* w_packed, const float* bias, void* y,
                       int relu, int out_f32, float* workspace, hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 0, "vc_conv_fwd_splitk");
  if (rc) return rc;
  VC_CHECK_ARG(x && w_packed && y && workspace, "vc_conv_fwd_splitk: null pointer");
  const int S = ksplit_plan(g, dtype);
  VC_CHECK_SHAPE(S > 1, "vc_conv_fwd_splitk: this geometry is not split (ask vc_conv_fwd_splitk_plan)");
  g.ksplit = S;
  const int units = ceil_div(g.M, BM) * ceil_div(g.Co, BN) * S;
  const int grid = units < nt_persist_blocks() ? units : nt_persist_blocks();
  launch_persist<0, 128, EPI_NONE>(g, grid, stream, x, w_packed, nullptr, workspace, 0, 1, 0, units, nullptr, nullptr,
                                   nullptr);
  VC_LAUNCH_CHECK("vc_conv_fwd_splitk");
  const int64_t n4 = (int64_t)g.M * g.Co / 4;
  const unsigned rgrid = (unsigned)(ceil_div64(n4, 256) < 4096 ? ceil_div64(n4, 256) : 4096);
  if (out_f32)
    hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3(rgrid), dim3(256), 0, stream, workspace, bias, (float*)y, n4,
                       g.Co / 4, S, (int64_t)g.M * g.Co, relu);
  else
    hipLaunchKernelGGL((splitk_reduce_kernel<bf16>), dim3(rgrid), dim3(256), 0, stream, workspace, bias, (bf16*)y, n4,
                       g.Co / 4, S, (int64_t)g.M * g.Co, relu);
  VC_LAUNCH_CHECK("vc_conv_fwd_splitk");
  return VC_OK;
}

int vc_conv_fwd_stats_supported(const int* desc, int dtype, int ldx, int out_f32) {
  ConvGeom g;
  if (fill_geom(g, desc, 0, "vc_conv_fwd_stats_supported")) return 0;
  if (ldx > 0) g.ldi = ldx;
  return (nt_persist_eligible(g, dtype) && !out_f32) ? 1 : 0;
}

int vc_conv_fwd_stats(const int* desc, int dtype, const void* x, int ldx, const void* w_packed, const float* bias,
                      void* y, int ldy, int relu, float* bn_partial, hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 0, "vc_conv_fwd_stats");
  if (rc) return rc;
  VC_CHECK_ARG(x && w_packed && y && bn_partial, "vc_conv_fwd_stats: null pointer");
  if (ldy > 0) g.ldc = ldy;
  if (ldx > 0) g.ldi = ldx;
  VC_CHECK_SHAPE(g.ldc >= g.Co && g.ldi >= g.Ci, "vc_conv_fwd_stats: leading dimension smaller than channel count");
  return launch_nt<0>(g, dtype, x, w_packed, bias, y, relu, 0, 0, stream, "vc_conv_fwd_stats", bn_partial);
}

int vc_conv_dgrad(const int* desc, int dtype, const void* dy, int lddy, const void* w_packed_t, void* dx, int lddx,
                  int out_f32, int beta, hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 1, "vc_conv_dgrad");
  if (rc) return rc;
  VC_CHECK_ARG(dy && w_packed_t && dx, "vc_conv_dgrad: null pointer");
  if (lddx > 0) g.ldc = lddx;
  if (lddy > 0) g.ldi = lddy;
  VC_CHECK_SHAPE(g.ldc >= g.Co && g.ldi >= g.Ci, "vc_conv_dgrad: leading dimension smaller than channel count");
  VC_CHECK_SHAPE(dtype != VC_BF16 || g.ldi % 8 == 0, "vc_conv_dgrad: bf16 path needs lddy %% 8 == 0");
  if (dtype == VC_BF16 && g.sd == 1 && (g.sh > 1 || g.sw > 1)) {
    // one GEMM per output phase (h mod sh, w mod sw): its reduction holds exactly the taps that reach the phase
    for (int a = 0; a < g.sh && a < g.Ho; ++a) {
      for (int b = 0; b < g.sw && b < g.Wo; ++b) {
        ConvGeom c = g;
        const int rh0 = (a + g.ph) % g.sh, rw0 = (b + g.pw) % g.sw;
        const int nth = rh0 < g.kh ? (g.kh - rh0 + g.sh - 1) / g.sh : 0;
        const int ntw = rw0 < g.kw ? (g.kw - rw0 + g.sw - 1) / g.sw : 0;
        if (nth * ntw == 0 && beta) continue;  // nothing reaches this phase and dx keeps what it holds
        const int Hc = (g.Ho - a + g.sh - 1) / g.sh, Wc = (g.Wo - b + g.sw - 1) / g.sw;
        c.oHo = g.Ho; c.oWo = g.Wo;
        c.o_h0 = a; c.o_hs = g.sh; c.o_w0 = b; c.o_ws = g.sw;
        c.wt_kh = g.kh; c.wt_kw = g.kw;
        c.wt_h0 = rh0; c.wt_hs = g.sh; c.wt_w0 = rw0; c.wt_ws = g.sw;
        c.Ho = Hc; c.Wo = Wc;
        c.kh = nth; c.kw = ntw;
        c.sh = c.sw = 1;
        c.ph = (a + g.ph) / g.sh; c.pw = (b + g.pw) / g.sw;
        c.K = g.kd * nth * ntw * g.Ci;  // 0: the phase is written with zeros
        c.M = g.N * g.Do * Hc * Wc;
        c.fWo = make_fastdiv(Wc);
        c.fHo = make_fastdiv(Hc);
        c.fkw = make_fastdiv(ntw > 0 ? ntw : 1);
        c.fkh = make_fastdiv(nth > 0 ? nth : 1);
        rc = launch_nt<2>(c, dtype, dy, w_packed_t, nullptr, dx, 0, out_f32, beta, stream, "vc_conv_dgrad");
        if (rc) return rc;
      }
    }
    return VC_OK;
  }
  return launch_nt<1>(g, dtype, dy, w_packed_t, nullptr, dx, 0, out_f32, beta, stream, "vc_conv_dgrad");
}

int vc_conv_dgrad_bn_tail_supported(const int* desc, int dtype) {
  ConvGeom g;
  if (fill_geom(g, desc, 1, "vc_conv_dgrad_bn_tail_supported")) return 0;
  return dgrad_tail_eligible(g, dtype) ? 1 : 0;
}

// vc_conv_dgrad (bf16, stride 1, dense rows) for the LAST contribution to the gradient of a residual block's
// output out = relu(bn(y) + shortcut): dx (+)= the data gradient, then dx *= the ReLU decisions in bits
// ([M][ci/8], from vc_bn_apply_mask), and bn_partial [ceil(M/64)][2][ci] receives per 64-row slab the sums of
// dz = dx and of dz * y (every entry written), from which vc_bn_bwd_sums_from_partial forms the BatchNorm
// backward reductions without a pass over dx and y.
int vc_conv_dgrad_bn_tail(const int* desc, int dtype, const void* dy, const void* w_packed_t, void* dx, int beta,
                          const void* tail_y, const uint8_t* tail_bits, float* bn_partial, hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 1, "vc_conv_dgrad_bn_tail");
  if (rc) return rc;
  VC_CHECK_ARG(dy && w_packed_t && dx && tail_y && tail_bits && bn_partial, "vc_conv_dgrad_bn_tail: null pointer");
  VC_CHECK_SHAPE(dgrad_tail_eligible(g, dtype),
                 "vc_conv_dgrad_bn_tail: unsupported geometry (ask vc_conv_dgrad_bn_tail_supported)");
  if (vc_nt8_eligible(g, 1, 0, EPI_BN_TAIL)) {
    vc_nt8_launch(g, 1, EPI_BN_TAIL, stream, dy, w_packed_t, nullptr, dx, 0, beta, bn_partial, tail_y, tail_bits);
    VC_LAUNCH_CHECK("vc_conv_dgrad_bn_tail");
    return VC_OK;
  }
  const int tiles = ceil_div(g.M, BM) * ceil_div(g.Co, BN);
  const int grid = tiles < nt_persist_blocks() ? tiles : nt_persist_blocks();
  launch_persist<1, 128, EPI_BN_TAIL>(g, grid, stream, dy, w_packed_t, nullptr, dx, 0, 0, beta, tiles, bn_partial, tail_y,
                                      tail_bits);
  VC_LAUNCH_CHECK("vc_conv_dgrad_bn_tail");
  return VC_OK;
}

// Stride-1 bf16 data gradient whose result is the gradient of relu(bn(y)) with y of the result's shape: the epilogue
// masks it with the ReLU decisions recomputed from y and the BatchNorm's scale / shift (stat = [4][G][ci]: mean,
// invstd, scale, shift; group = rows_per_group consecutive rows, a multiple of 64) and writes bn_partial
// [M/64][2][ci] = per 64-row slab the sums of dz and dz * y, from which vc_bn_bwd_sums_from_partial forms the
// BatchNorm backward reductions: the separate reduction pass over (dx, y) is not needed.
int vc_conv_dgrad_bn_pre_supported(const int* desc, int dtype, int rows_per_group) {
  ConvGeom g;
  if (fill_geom(g, desc, 1, "vc_conv_dgrad_bn_pre_supported")) return 0;
  return (nt_persist_eligible(g, dtype) && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.Co % 8 == 0 && g.ldc == g.Co &&
          rows_per_group > 0 && rows_per_group % 64 == 0 && g.M % rows_per_group == 0) ? 1 : 0;
}

int vc_conv_dgrad_bn_pre(const int* desc, int dtype, const void* dy, const void* w_packed_t, void* dx, const void* y,
                         const float* stat, int rows_per_group, float* bn_partial, hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 1, "vc_conv_dgrad_bn_pre");
  if (rc) return rc;
  VC_CHECK_ARG(dy && w_packed_t && dx && y && stat && bn_partial, "vc_conv_dgrad_bn_pre: null pointer");
  VC_CHECK_SHAPE(vc_conv_dgrad_bn_pre_supported(desc, dtype, rows_per_group) == 1,
                 "vc_conv_dgrad_bn_pre: unsupported geometry (ask vc_conv_dgrad_bn_pre_supported)");
  if (vc_nt8_eligible(g, 1, 0, EPI_BN_PRE)) {
    vc_nt8_launch(g, 1, EPI_BN_PRE, stream, dy, w_packed_t, stat, dx, rows_per_group, 0, bn_partial, y, nullptr);
  } else if (g.Co <= nt_narrow() && nt_tall() && bl_eligible(g, 1, true)) {
    const int t = ceil_div(g.M, 256) * ceil_div(g.Co, 64);
    const int grid = t < nt_persist_blocks() ? t : nt_persist_blocks();
    hipLaunchKernelGGL((igemm_nt_persist_kernel<1, 64, EPI_BN_PRE, true, 2, 256>), dim3(grid), dim3(NTHR), 0, stream, g,
                       (const bf16*)dy, (const bf16*)w_packed_t, stat, dx, rows_per_group, 0, 0, t, bn_partial,
                       (const bf16*)y, (const uint8_t*)nullptr);
  } else if (g.Co <= nt_narrow()) {
    const int t64 = ceil_div(g.M, BM) * ceil_div(g.Co, 64);
    const int grid = t64 < nt_persist_blocks() * 3 / 2 ? t64 : nt_persist_blocks() * 3 / 2;
    launch_persist<1, 64, EPI_BN_PRE>(g, grid, stream, dy, w_packed_t, stat, dx, rows_per_group, 0, 0, t64, bn_partial,
                                      y, nullptr);
  } else {
    const int tiles = ceil_div(g.M, BM) * ceil_div(g.Co, BN);
    const int grid = tiles < nt_persist_blocks() ? tiles : nt_persist_blocks();
    launch_persist<1, 128, EPI_BN_PRE>(g, grid, stream, dy, w_packed_t, stat, dx, rows_per_group, 0, 0, tiles,
                                       bn_partial, y, nullptr);
  }
  VC_LAUNCH_CHECK("vc_conv_dgrad_bn_pre");
  return VC_OK;
}

// Geometries whose epilogue can take a side tensor of the output's layout (EPI_RESIDUAL / EPI_GELU_BWD)
int vc_conv_epilogue_side_supported(const int* desc, int dtype, int mode) {
  ConvGeom g;
  if (fill_geom(g, desc, mode ? 1 : 0, "vc_conv_epilogue_side_supported")) return 0;
  return dgrad_tail_eligible(g, dtype) ? 1 : 0;
}

// y = row_scale * (conv(x, w) + bias) + residual, residual and y of the same [M][co] layout, written out of place
// (bf16).  Replaces the x + drop_path(f(x)) of the Swin blocks (timm SwinTransformerBlock behind
// model_components.py:254, 412).  row_scale (nullable = all ones): fp32, one value per rows_per_sample consecutive
// rows (stochastic depth: 0 or 1 / keep probability per frame, from vc_drop_path_scales).
int vc_conv_fwd_residual(const int* desc, int dtype, const void* x, const void* w_packed, const float* bias,
                         const void* residual, const float* row_scale, int rows_per_sample, void* y,
                         hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 0, "vc_conv_fwd_residual");
  if (rc) return rc;
  VC_CHECK_ARG(x && w_packed && residual && y, "vc_conv_fwd_residual: null pointer");
  VC_CHECK_SHAPE(dgrad_tail_eligible(g, dtype),
                 "vc_conv_fwd_residual: unsupported geometry (ask vc_conv_epilogue_side_supported)");
  VC_CHECK_SHAPE(!row_scale || (rows_per_sample > 0 && g.M % rows_per_sample == 0),
                 "vc_conv_fwd_residual: %d rows do not split into samples of %d", g.M, rows_per_sample);
  const int tiles = ceil_div(g.M, BM) * ceil_div(g.Co, BN);
  const int grid = tiles < nt_persist_blocks() ? tiles : nt_persist_blocks();
  launch_persist<0, 128, EPI_RESIDUAL>(g, grid, stream, x, w_packed, bias, y, row_scale ? rows_per_sample : 0, 0, 0,
                                       tiles, nullptr, residual, reinterpret_cast<const uint8_t*>(row_scale));
  VC_LAUNCH_CHECK("vc_conv_fwd_residual");
  return VC_OK;
}

// y = relu(conv(x) + bias + residual): the tail of a residual block whose BatchNorm has been folded into the
// convolution (eval mode).  Same geometries as vc_conv_fwd_residual.
int vc_conv_fwd_residual_relu(const int* desc, int dtype, const void* x, const void* w_packed, const float* bias,
                              const void* residual, void* y, hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 0, "vc_conv_fwd_residual_relu");
  if (rc) return rc;
  VC_CHECK_ARG(x && w_packed && residual && y, "vc_conv_fwd_residual_relu: null pointer");
  VC_CHECK_SHAPE(dgrad_tail_eligible(g, dtype),
                 "vc_conv_fwd_residual_relu: unsupported geometry (ask vc_conv_epilogue_side_supported)");
  const int tiles = ceil_div(g.M, BM) * ceil_div(g.Co, BN);
  const int grid = tiles < nt_persist_blocks() ? tiles : nt_persist_blocks();
  launch_persist<0, 128, EPI_RESIDUAL>(g, grid, stream, x, w_packed, bias, y, -1, 0, 0, tiles, nullptr, residual, nullptr);
  VC_LAUNCH_CHECK("vc_conv_fwd_residual_relu");
  return VC_OK;
}

// y = conv(x) + bias and y_gelu = gelu(y) (of the value as stored in y) in one pass: the activation between the two
// linear layers of an MLP without a pass of its own.  Geometries: vc_conv_epilogue_side_supported(desc, dtype, 0).
int vc_conv_fwd_gelu_dual(const int* desc, int dtype, const void* x, const void* w_packed, const float* bias, void* y,
                          void* y_gelu, hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 0, "vc_conv_fwd_gelu_dual");
  if (rc) return rc;
  VC_CHECK_ARG(x && w_packed && y && y_gelu, "vc_conv_fwd_gelu_dual: null pointer");
  VC_CHECK_SHAPE(dgrad_tail_eligible(g, dtype),
                 "vc_conv_fwd_gelu_dual: unsupported geometry (ask vc_conv_epilogue_side_supported)");
  const int tiles = ceil_div(g.M, BM) * ceil_div(g.Co, BN);
  const int grid = tiles < nt_persist_blocks() ? tiles : nt_persist_blocks();
  launch_persist<0, 128, EPI_GELU_DUAL>(g, grid, stream, x, w_packed, bias, y, 0, 0, 0, tiles, nullptr, y_gelu, nullptr);
  VC_LAUNCH_CHECK("vc_conv_fwd_gelu_dual");
  return VC_OK;
}

// dx = (data gradient of conv) * gelu'(pre), pre = the GELU's input of dx's own layout (bf16): the gradient of
// gelu(fc1(x)) w.r.t. fc1's output, straight from the data-gradient GEMM of the layer that follows the GELU.
int vc_conv_dgrad_gelu(const int* desc, int dtype, const void* dy, const void* w_packed_t, const void* pre, void* dx,
                       hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 1, "vc_conv_dgrad_gelu");
  if (rc) return rc;
  VC_CHECK_ARG(dy && w_packed_t && pre && dx, "vc_conv_dgrad_gelu: null pointer");
  VC_CHECK_SHAPE(dgrad_tail_eligible(g, dtype),
                 "vc_conv_dgrad_gelu: unsupported geometry (ask vc_conv_epilogue_side_supported)");
  const int tiles = ceil_div(g.M, BM) * ceil_div(g.Co, BN);
  const int grid = tiles < nt_persist_blocks() ? tiles : nt_persist_blocks();
  launch_persist<1, 128, EPI_GELU_BWD>(g, grid, stream, dy, w_packed_t, nullptr, dx, 0, 0, 0, tiles, nullptr, pre, nullptr);
  VC_LAUNCH_CHECK("vc_conv_dgrad_gelu");
  return VC_OK;
}

extern "C" int vc_colsum(int dtype, const void* a, float* out, int M, int C, int lda, hipStream_t stream);

int vc_conv_wgrad(const int* desc, int dtype, const void* x, int ldx, const void* dy, int lddy, float* dw_packed,
                  float* dbias, int force_naive, hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, 0, "vc_conv_wgrad");
  if (rc) return rc;
  VC_CHECK_ARG(x && dy && dw_packed, "vc_conv_wgrad: null pointer");
  if (ldx > 0) g.ldi = ldx;
  const int ldy_ = lddy > 0 ? lddy : g.Co;
  VC_CHECK_SHAPE(g.ldi >= g.Ci && ldy_ >= g.Co, "vc_conv_wgrad: leading dimension smaller than channel count");
  VC_CHECK_SHAPE(dtype != VC_BF16 || (g.ldi % 8 == 0 && ldy_ % 8 == 0), "vc_conv_wgrad: bf16 path needs ld %% 8 == 0");
  if (dtype == VC_BF16 && !force_naive) {
    VC_CHECK_SHAPE(g.Ci % 8 == 0 && g.Co % 8 == 0, "vc_conv_wgrad: bf16 path needs Ci %% 8 == 0 and Co %% 8 == 0");
    const bool fast = use_glds() && !use_generic_nt();
    if (fast && vc_wgrad8_eligible(g, ldy_, dbias != nullptr)) {  // wide layers: 256 x 256 tiles, one workgroup per CU
      const size_t nw = (size_t)g.Co * g.K;
      const bool det = vc_deterministic();
      int splits, mps;
      vc_wgrad8_plan(g, vc_det_capacity(), det, &splits, &mps);
      const bool ordered = det && splits > 1;
      float* target = dw_packed;
      if (ordered) {
        target = vc_det_workspace((size_t)splits * nw, stream, "vc_conv_wgrad");
        if (!target) return VC_ERR_ARG;
      }
      float* const db_fused = det ? nullptr : dbias;  // deterministic mode: an ordered column-sum pass below
      vc_wgrad8_launch(g, x, dy, ldy_, target, splits, mps, ordered ? nw : 0, db_fused, stream);
      VC_LAUNCH_CHECK("vc_conv_wgrad");
      if (ordered)
        if (int rc2 = vc_det_reduce(dw_packed, target, nw, splits, stream)) return rc2;
      if (dbias && !db_fused) return vc_colsum(dtype, dy, dbias, g.M, g.Co, ldy_, stream);
      return VC_OK;
    }
    float* dbias_fused = fast ? dbias : nullptr;
    int tco = (fast && g.Co <= 64 && wgrad_narrow()) ? 64 : 128;
    int tcol = (fast && g.K <= 64 && wgrad_narrow()) ? 64 : 128;
    // measured on the ResNet-50 shapes: short pixel ranges (late stages) and wide 1x1 layers run 5-15% faster
    // with three narrower blocks per CU than with two 128x128 ones
    const bool taps1 = g.kd * g.kh * g.kw == 1;
    if (fast && wgrad_narrow() == 1 && tco == 128 && tcol == 128 &&
        (g.M <= 131072 || (taps1 && (g.Ci >= 512 || g.Co >= 512))))
      tcol = 64;
    if (fast && wgrad_narrow() == 2) tco = 64;   // tuning runs: force the narrow forms
    if (fast && wgrad_narrow() == 3) tcol = 64;
    const int tiles_co = ceil_div(g.Co, tco), tiles_col = ceil_div(g.K, tcol);
    const int ntiles = tiles_co * tiles_col;
    int splits = ceil_div(wgrad_target_blocks() * (tco + tcol < 256 ? 3 : 2) / 2, ntiles);
    const int max_splits = ceil_div(g.M, BK);
    if (splits > max_splits) splits = max_splits;
    // Every split adds the whole fp32 weight tile with atomics (about 1.3 TB/s chip-wide): a split must bring
    // enough rows for its products to outweigh that, which depends on the row count alone (both scale with Co*K).
    // Matters for short, wide layers (Swin stages 3-4, ResNet layer4): fewer, longer workgroups win there.
    const int split_cap = g.M / wgrad_min_rows_per_split() > 2 ? g.M / wgrad_min_rows_per_split() : 2;
    if (splits > split_cap) splits = split_cap;
    if (splits < 1) splits = 1;
    // deterministic mode: every split adds into its own copy of dw (and dbias) in the workspace and the copies are
    // added in split order afterwards; as many splits as the workspace holds copies (one split needs none)
    const size_t nw = (size_t)g.Co * g.K, per_split = nw + (dbias_fused ? (size_t)g.Co : 0);
    const bool det = vc_deterministic();
    if (det && (size_t)splits * per_split > vc_det_capacity()) {
      splits = (int)(vc_det_capacity() / per_split);
      if (splits < 1) splits = 1;
    }
    int mps = ceil_div(ceil_div(g.M, splits), BK) * BK;
    splits = ceil_div(g.M, mps);
    const dim3 grid(ntiles * splits);
    const bool ordered = det && splits > 1;
    float* const dw_user = dw_packed;
    float* const db_user = dbias_fused;
    size_t split_stride = 0;
    if (ordered) {
      float* ws = vc_det_workspace((size_t)splits * per_split, stream, "vc_conv_wgrad");
      if (!ws) return VC_ERR_ARG;
      dw_packed = ws;
      if (dbias_fused) dbias_fused = ws + (size_t)splits * nw;
      split_stride = nw;
    }
    if (fast && tco == 64 && tcol == 64)
      hipLaunchKernelGGL((igemm_tn_wgrad_fast_kernel<64, 64>), grid, dim3(NTHR), 0, stream, g, (const bf16*)x,
                         (const bf16*)dy, ldy_, dw_packed, mps, ntiles, tiles_col, dbias_fused, split_stride);
    else if (fast && tco == 64)
      hipLaunchKernelGGL((igemm_tn_wgrad_fast_kernel<64, 128>), grid, dim3(NTHR), 0, stream, g, (const bf16*)x,
                         (const bf16*)dy, ldy_, dw_packed, mps, ntiles, tiles_col, dbias_fused, split_stride);
    else if (fast && tcol == 64)
      hipLaunchKernelGGL((igemm_tn_wgrad_fast_kernel<128, 64>), grid, dim3(NTHR), 0, stream, g, (const bf16*)x,
                         (const bf16*)dy, ldy_, dw_packed, mps, ntiles, tiles_col, dbias_fused, split_stride);
    else if (fast)
      hipLaunchKernelGGL((igemm_tn_wgrad_fast_kernel<128, 128>), grid, dim3(NTHR), 0, stream, g, (const bf16*)x,
                         (const bf16*)dy, ldy_, dw_packed, mps, ntiles, tiles_col, dbias_fused, split_stride);
    else if (use_glds())
      hipLaunchKernelGGL((igemm_tn_wgrad_bf16_kernel<true>), dim3(ntiles * splits), dim3(NTHR), 0, stream, g,
                         (const bf16*)x, (const bf16*)dy, ldy_, dw_packed, mps, ntiles, tiles_col, split_stride);
    else
      hipLaunchKernelGGL((igemm_tn_wgrad_bf16_kernel<false>), dim3(ntiles * splits), dim3(NTHR), 0, stream, g,
                         (const bf16*)x, (const bf16*)dy, ldy_, dw_packed, mps, ntiles, tiles_col, split_stride);
    if (ordered) {
      VC_LAUNCH_CHECK("vc_conv_wgrad");
      if (int rc2 = vc_det_reduce(dw_user, dw_packed, nw, splits, stream)) return rc2;
      if (db_user)
        if (int rc2 = vc_det_reduce(db_user, dbias_fused, (size_t)g.Co, splits, stream)) return rc2;
      dw_packed = dw_user;
    }
  } else if (dtype == VC_BF16) {
    const int64_t total = (int64_t)g.Co * g.K;
    hipLaunchKernelGGL((wgrad_naive_kernel<bf16>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream, g,
                       (const bf16*)x, (const bf16*)dy, ldy_, dw_packed);
  } else if (dtype == VC_F32) {
    const int64_t total = (int64_t)g.Co * g.K;
    hipLaunchKernelGGL((wgrad_naive_kernel<float>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream, g,
                       (const float*)x, (const float*)dy, ldy_, dw_packed);
  } else {
    vc_set_error("vc_conv_wgrad: unsupported dtype %d", dtype);
    return VC_ERR_DTYPE;
  }
  VC_LAUNCH_CHECK("vc_conv_wgrad");
  // bias gradient: fused into the MFMA kernel above where that ran, a column-sum pass over dy otherwise
  const bool fused = dtype == VC_BF16 && !force_naive && use_glds() && !use_generic_nt();
  if (dbias && !fused) return vc_colsum(dtype, dy, dbias, g.M, g.Co, ldy_, stream);
  return VC_OK;
}

// Cross-check entry: the naive kernel on bf16 data (used by the tests to validate the MFMA path).
int vc_conv_fwd_naive_bf16(const int* desc, int mode, const void* src, const void* w_packed, const float* bias,
                           void* out, int relu, int out_f32, hipStream_t stream) {
  ConvGeom g;
  int rc = fill_geom(g, desc, mode, "vc_conv_fwd_naive_bf16");
  if (rc) return rc;
  const int64_t total = (int64_t)g.M * g.Co;
  if (mode == 0)
    hipLaunchKernelGGL((conv_naive_kernel<bf16, 0>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream, g,
                       (const bf16*)src, (const bf16*)w_packed, bias, out, relu, out_f32, 0);
  else
    hipLaunchKernelGGL((conv_naive_kernel<bf16, 1>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream, g,
                       (const bf16*)src, (const bf16*)w_packed, bias, out, relu, out_f32, 0);
  VC_LAUNCH_CHECK("vc_conv_fwd_naive_bf16");
  return VC_OK;
}

int vc_pack_weight(int dtype, const float* w, void* w_fwd, void* w_t, int co, int ci, int taps, int ci_pad, int co_pad,
                   hipStream_t stream) {
  VC_CHECK_ARG(w && (w_fwd || w_t), "vc_pack_weight: null pointer");
  VC_CHECK_SHAPE(co > 0 && ci > 0 && taps > 0 && ci_pad >= ci && co_pad >= co, "vc_pack_weight: bad shape");
  const int64_t nf = w_fwd ? (int64_t)co * taps * ci_pad : 0, nt = w_t ? (int64_t)ci * taps * co_pad : 0;
  const int64_t total = nf > nt ? nf : nt;
  if (w_fwd && !w_t && relayout_tiled(co, ci_pad, taps) && (dtype == VC_BF16 || dtype == VC_F32)) {
    const dim3 grid(ceil_div(taps, TT), ceil_div(ci_pad, TT), co);
    if (dtype == VC_BF16)
      hipLaunchKernelGGL((pack_weight_tiled_kernel<bf16>), grid, dim3(256), 0, stream, w, (bf16*)w_fwd, ci, taps, ci_pad);
    else
      hipLaunchKernelGGL((pack_weight_tiled_kernel<float>), grid, dim3(256), 0, stream, w, (float*)w_fwd, ci, taps,
                         ci_pad);
    VC_LAUNCH_CHECK("vc_pack_weight");
    return VC_OK;
  }
  if (dtype == VC_BF16)
    hipLaunchKernelGGL((pack_weight_kernel<bf16>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream, w,
                       (bf16*)w_fwd, (bf16*)w_t, co, ci, taps, ci_pad, co_pad);
  else if (dtype == VC_F32)
    hipLaunchKernelGGL((pack_weight_kernel<float>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream, w,
                       (float*)w_fwd, (float*)w_t, co, ci, taps, ci_pad, co_pad);
  else {
    vc_set_error("vc_pack_weight: unsupported dtype %d", dtype);
    return VC_ERR_DTYPE;
  }
  VC_LAUNCH_CHECK("vc_pack_weight");
  return VC_OK;
}

int vc_pack_weight_flat_t(int dtype, const float* w, void* w_t, int co, int ci, int taps, int ci_pad, int co_pad,
                          hipStream_t stream) {
  VC_CHECK_ARG(w && w_t, "vc_pack_weight_flat_t: null pointer");
  VC_CHECK_SHAPE(co > 0 && ci > 0 && taps > 0 && ci_pad >= ci && co_pad >= co, "vc_pack_weight_flat_t: bad shape");
  const int64_t total = (int64_t)taps * ci_pad * co_pad;
  if (relayout_tiled(ci_pad, co_pad, taps) && (dtype == VC_BF16 || dtype == VC_F32)) {
    const dim3 grid(ceil_div(taps, TT), ceil_div(co_pad, TT), ci_pad);
    if (dtype == VC_BF16)
      hipLaunchKernelGGL((pack_weight_flat_t_tiled_kernel<bf16>), grid, dim3(256), 0, stream, w, (bf16*)w_t, co, ci, taps,
                         ci_pad, co_pad);
    else
      hipLaunchKernelGGL((pack_weight_flat_t_tiled_kernel<float>), grid, dim3(256), 0, stream, w, (float*)w_t, co, ci,
                         taps, ci_pad, co_pad);
    VC_LAUNCH_CHECK("vc_pack_weight_flat_t");
    return VC_OK;
  }
  if (dtype == VC_BF16)
    hipLaunchKernelGGL((pack_weight_flat_t_kernel<bf16>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream,
                       w, (bf16*)w_t, co, ci, taps, ci_pad, co_pad);
  else if (dtype == VC_F32)
    hipLaunchKernelGGL((pack_weight_flat_t_kernel<float>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0,
                       stream, w, (float*)w_t, co, ci, taps, ci_pad, co_pad);
  else {
    vc_set_error("vc_pack_weight_flat_t: unsupported dtype %d", dtype);
    return VC_ERR_DTYPE;
  }
  VC_LAUNCH_CHECK("vc_pack_weight_flat_t");
  return VC_OK;
}

int vc_pack_weight_unfold(int dtype, const float* w, void* w_fwd, int co, int ci, int taps_outer, int kw, int jp,
                          int co_pad, hipStream_t stream) {
  VC_CHECK_ARG(w && w_fwd, "vc_pack_weight_unfold: null pointer");
  VC_CHECK_SHAPE(co > 0 && ci > 0 && taps_outer > 0 && kw > 0 && jp >= kw * ci && co_pad >= co,
                 "vc_pack_weight_unfold: bad shape");
  const int64_t total = (int64_t)co_pad * taps_outer * jp;
  if (dtype == VC_BF16)
    hipLaunchKernelGGL((pack_weight_unfold_kernel<bf16>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream,
                       w, (bf16*)w_fwd, co, ci, taps_outer, kw, jp, co_pad);
  else if (dtype == VC_F32)
    hipLaunchKernelGGL((pack_weight_unfold_kernel<float>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0,
                       stream, w, (float*)w_fwd, co, ci, taps_outer, kw, jp, co_pad);
  else {
    vc_set_error("vc_pack_weight_unfold: unsupported dtype %d", dtype);
    return VC_ERR_DTYPE;
  }
  VC_LAUNCH_CHECK("vc_pack_weight_unfold");
  return VC_OK;
}

int vc_unpack_wgrad_unfold(float* dw_packed, float* dw, int co, int ci, int taps_outer, int kw, int jp, int flags,
                           hipStream_t stream) {
  VC_CHECK_ARG(dw_packed && dw, "vc_unpack_wgrad_unfold: null pointer");
  const int64_t total = (int64_t)co * ci * taps_outer * kw;
  hipLaunchKernelGGL(unpack_wgrad_unfold_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream,
                     dw_packed, dw, co, ci, taps_outer, kw, jp, flags);
  VC_LAUNCH_CHECK("vc_unpack_wgrad_unfold");
  return VC_OK;
}

int vc_unpack_wgrad(float* dw_packed, float* dw, int co, int ci, int taps, int ci_pad, int flags,
                    hipStream_t stream) {
  VC_CHECK_ARG(dw_packed && dw, "vc_unpack_wgrad: null pointer");
  const int64_t total = (int64_t)co * ci * taps;
  if (relayout_tiled(co, ci, taps)) {
    hipLaunchKernelGGL(unpack_wgrad_tiled_kernel, dim3(ceil_div(taps, TT), ceil_div(ci, TT), co), dim3(256), 0, stream,
                       dw_packed, dw, ci, taps, ci_pad, flags);
    VC_LAUNCH_CHECK("vc_unpack_wgrad");
    return VC_OK;
  }
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream, dw_packed, dw,
                     co, ci, taps, ci_pad, flags);
  VC_LAUNCH_CHECK("vc_unpack_wgrad");
  return VC_OK;
}

}  // extern "C"
