// Implicit-GEMM NT kernel (forward / data gradient / linear) on 256-pixel x 256-channel output tiles, gfx950.
//
// One workgroup of 8 waves per CU (two waves per SIMD, 256 registers each).  The waves form a 2 x 4 grid, each owning
// 128 pixels x 64 channels of the tile = 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16.  A 64-deep k-tile is
// worked off in four phases of 16 products (one quadrant of the wave's tile: 64 pixels x 32 channels); a phase is
//     fragment reads of the quadrant's new operand rows + LDS-DMA requests for ONE staging unit
//     s_barrier, wait for the fragments, 16 products, s_barrier
// and the two wave rows run one barrier apart, so that on every SIMD one wave multiplies while its partner reads and
// requests.  Staging units (16 KiB = 128 rows of one operand = two 1 KiB requests per wave) are re-requested two
// phases after their last read, four units are in flight at any time and waits are counted (vmcnt(8)): loads stay in
// flight across barriers and across k-tiles; nothing in the loop drains them.
//
// The gather is the buffer-load form of the 128-wide persistent kernel (vc_igemm.hip): per staged row one 32-bit byte
// offset of its window origin and a bit mask of the filter taps that fall inside the source, both computed once per
// output tile; lanes whose tap is padding (or whose row lies beyond the matrix) carry an offset beyond the
// descriptor's range, for which the hardware writes zeros to LDS.
//
// Persistent: a workgroup walks output tiles v, v + G, ...; the staging stream runs through the tile boundaries except
// that the two units which would land in the stage the epilogue parks its output image in are requested after the
// epilogue.  The epilogues are those of the 128-wide kernel (nt_epilogue_lds), called once per 64-pixel half of the
// wave's tile.
#include <stdlib.h>

#include "vc_igemm_dev.h"

namespace {

constexpr int BM8 = 256;   // pixels per tile
constexpr int NT8 = 512;   // threads per workgroup

template <int VAR>
__device__ __forceinline__ bf16x8 ldfrag(const bf16* p) {
  if (VAR & 2) {  // ablation: no LDS read
    bf16x8 r;
    asm volatile("" : "=v"(r));
    return r;
  }
  return *reinterpret_cast<const bf16x8*>(p);
}

template <int VAR>
__device__ __forceinline__ f32x4 vmfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if (VAR & 16) {  // ablation: no product, operands and accumulator kept alive
    f32x4 r = c;
    asm volatile("" : "+v"(r) : "v"(a), "v"(b));
    return r;
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int MODE, int EPI, int VAR = 0>
__global__ __launch_bounds__(NT8, 2) void igemm_nt8_kernel(ConvGeom g, const bf16* __restrict__ src,
                                                           const bf16* __restrict__ wgt, const float* __restrict__ bias,
                                                           bf16* __restrict__ out, int relu, int beta, int ntiles,
                                                           float* __restrict__ bn_partial,
                                                           const bf16* __restrict__ tail_y,
                                                           const uint8_t* __restrict__ tail_bits) {
  constexpr int TN = 256;
  constexpr int ASZ = BM8 * BK, WSZ = TN * BK, STG = ASZ + WSZ;  // elements per stage
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * STG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 2, wc = wave & 3;  // wave row (pixel half), wave column (channel quarter)
  const int tiles_n = (g.Co + TN - 1) / TN;
  int vblock;
  xcd_remap(vblock);
  const int G = gridDim.x;

  // staging role: LDS row lrow (+ unit offsets), physical 16-byte slot pchunk holding logical k-chunk lchunk
  const int lrow = tid >> 3, pchunk = tid & 7;
  const int lchunk = pchunk ^ ((lrow >> 1) & 7);

  const int ncb = (g.Ci + BK - 1) / BK;      // k-tiles per tap
  const int nk = g.kd * g.kh * g.kw * ncb;   // k-tiles per output tile (>= 1)
  const bool ktail = (g.Ci % BK) != 0;       // single-tap layers only: chunks beyond Ci read zeros

  struct KCursor {
    int tile, kt, tap, cb, rd, rh, rw, st;
  };
  // k-tiles run channel block by channel block and, inside a block, tap by tap: the taps of a block re-read the same
  // source lines (shifted by a pixel or a row), which are then still in the L1 / L2 caches; tap-major order re-reads a
  // line only after the whole channel range has passed through (measured: the re-reads then come from beyond L2)
  const int ntaps = g.kd * g.kh * g.kw;
  auto advance = [&](KCursor& c) {
    ++c.tap;
    if (++c.rw == g.kw) {
      c.rw = 0;
      if (++c.rh == g.kh) {
        c.rh = 0;
        ++c.rd;
      }
    }
    if (c.tap == ntaps) {
      c.tap = c.rd = c.rh = c.rw = 0;
      ++c.cb;
    }
    if (++c.kt == nk) {
      c.kt = 0;
      c.tap = c.cb = c.rd = c.rh = c.rw = 0;
      c.tile += G;
    }
    c.st ^= 1;
  };
  KCursor cu{vblock, 0, 0, 0, 0, 0, 0, 0};

  constexpr unsigned SENT = 0xFFFFFFF0u;  // beyond every descriptor's range
  unsigned abase[4], amask[4], avo[4], wvo[4];  // pixel rows [h*2+g], weight rows [c*2+g]
  int s_kb = 0, s_wk = 0;                       // uniform: byte advance inside the tap, byte offset of the tap's weights
  bool s_dead = false;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(src), 0, (int)((unsigned)(g.N * g.Di * g.Hi * g.Wi) * (unsigned)g.ldi * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(wgt), 0, (int)((unsigned)g.Co * (unsigned)g.ldw * 2u), 0x00020000);

  // VAR & 1024 (timing experiment): the same requests as plain loads into a register nobody reads
  f32x4 dummy = {0.f, 0.f, 0.f, 0.f};
  auto DMA = [&](const __amdgpu_buffer_rsrc_t& rs, bf16* lds, int voff, int soff) {
    if (VAR & 1024) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dummy) : "v"(voff), "s"(rs), "s"(soff));
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, soff, 0, 0);
  };
  // unit A0 of the cursor's k-tile; the first unit of a k-tile also refreshes the row / tap state
  auto stage_a0 = [&]() {
    if (cu.kt == 0) {
      const int tile_m = cu.tile / tiles_n, n0 = (cu.tile - tile_m * tiles_n) * TN;
      const int m0 = tile_m * BM8;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const RowPos r = decode_row<MODE>(g, m0 + gg * 128 + h * 64 + lrow);  // rows beyond M: far outside
          abase[h * 2 + gg] =
              (unsigned)((((r.n * g.Di + r.d0) * g.Hi + r.h0) * g.Wi + r.w0) * g.ldi * 2 + lchunk * 16);
          amask[h * 2 + gg] = tap_mask<MODE>(g, r);
        }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int co = n0 + (gg * 2 + (lrow >> 5)) * 64 + c * 32 + (lrow & 31);
          wvo[c * 2 + gg] = co < g.Co ? (unsigned)(co * g.ldw * 2 + lchunk * 16) : SENT;
        }
    }
    if (cu.kt == 0 || ntaps > 1) {
      int toff = ((cu.rd * g.Hi + cu.rh) * g.Wi + cu.rw) * g.ldi * 2;
      if (MODE != 0) toff = -toff;
#pragma unroll
      for (int i = 0; i < 4; ++i) avo[i] = ((amask[i] >> cu.tap) & 1u) ? abase[i] + (unsigned)toff : SENT;
      s_wk = MODE == 2 ? ((cu.rd * g.wt_kh + g.wt_h0 + g.wt_hs * cu.rh) * g.wt_kw + g.wt_w0 + g.wt_ws * cu.rw) * g.Ci * 2
                       : cu.tap * g.Ci * 2;
    }
    s_kb = cu.cb * (BK * 2);
    s_dead = ktail && cu.cb * BK + lchunk * 8 >= g.Ci;
    bf16* sA = smem + cu.st * STG;
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
      DMA(rs_a, sA + (gg * 128 + 8 * wave) * BK, (int)(s_dead ? SENT : avo[gg]), s_kb);
  };
  auto stage_a1 = [&]() {  // the last unit of a k-tile: moves the cursor on
    bf16* sA = smem + cu.st * STG;
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
      DMA(rs_a, sA + (gg * 128 + 64 + 8 * wave) * BK, (int)(s_dead ? SENT : avo[2 + gg]), s_kb);
    advance(cu);
  };
  auto stage_w = [&](int c) {
    bf16* sW = smem + cu.st * STG + ASZ;
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
      DMA(rs_w, sW + ((gg * 2 + (wave >> 2)) * 64 + c * 32 + (wave & 3) * 8) * BK, (int)(s_dead ? SENT : wvo[c * 2 + gg]),
          s_wk + s_kb);
  };
  // counted wait: everything but the four newest units (8 requests) has landed; once the stream has run dry the
  // newest requests are no longer there to count on
  auto wait_stream = [&]() {
    if ((VAR & 32) && cu.tile < ntiles) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // timing experiment only
    else if ((VAR & 128) && cu.tile < ntiles) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (cu.tile < ntiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  f32x4 acc[2][4][4];  // [pixel half h][channel tile ci = c*2 + i][pixel tile j]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row r keeps its 16-byte chunks XORed with (r >> 1) & 7, the same value for every row
  // l15 + 16 * i one lane reads
  const int l15 = lane & 15, q = lane >> 4, x = (l15 >> 1) & 7;
  const int c0 = ((q ^ x) << 3), c1 = (((q ^ x) ^ 4) << 3);
  const int aoff = (wp * 128 + l15) * BK, woff = ASZ + (wc * 64 + l15) * BK;

  bf16x8 af[4][2], wf0[2][2], wf1[2][2];
  // VAR & 64: cycle stamps (diagnostic build).  Per phase: t0 = phase start, tw = before the counted wait, tr = before
  // the first barrier, tm = products start, te = products done, next t0 = after the second barrier.
  unsigned pa_issue = 0, pa_wait = 0, pa_b1 = 0, pa_mfma = 0, pa_b2 = 0, pa_n = 0, pa_epi = 0;
  unsigned long long t0 = 0, tw = 0, tr = 0, tm = 0, te = 0;
#define STAMP(x) do { if (VAR & 64) x = __builtin_readcyclecounter(); } while (0)
#define PHASE_END() do { if (VAR & 64) { unsigned long long tn = __builtin_readcyclecounter(); \
    pa_issue += (unsigned)(tw - t0); pa_wait += (unsigned)(tr - tw); pa_b1 += (unsigned)(tm - tr); \
    pa_mfma += (unsigned)(te - tm); pa_b2 += (unsigned)(tn - te); ++pa_n; t0 = tn; } } while (0)

  unsigned long long ck0 = 0, rt0 = 0;
  if (VAR & 2048) { ck0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  if constexpr ((VAR & 256) != 0) {
    // ---- free-running schedule: one barrier per k-tile, the next k-tile's requests spread between the quadrants;
    // the two waves of a SIMD are not held in step, so one multiplies while the other is held up issuing ----
    if (cu.tile < ntiles) {
      stage_a0();
      stage_w(0);
      stage_w(1);
      stage_a1();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int c_tile = vblock, c_kt = 0, cs = 0;
    while (c_tile < ntiles) {
      const bool last = c_kt == nk - 1;
      const bool more = cu.tile < ntiles;
      const bf16* tS = smem + cs * STG;
      if ((VAR & 512) && more) {
        stage_a0();
        stage_w(0);
        stage_w(1);
        stage_a1();
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wf0[i][0] = *reinterpret_cast<const bf16x8*>(tS + woff + c0 + i * 16 * BK);
        wf0[i][1] = *reinterpret_cast<const bf16x8*>(tS + woff + c1 + i * 16 * BK);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        af[j][0] = *reinterpret_cast<const bf16x8*>(tS + aoff + c0 + j * 16 * BK);
        af[j][1] = *reinterpret_cast<const bf16x8*>(tS + aoff + c1 + j * 16 * BK);
      }
      if (!(VAR & 512) && more) stage_a0();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wf1[i][0] = *reinterpret_cast<const bf16x8*>(tS + woff + c0 + (32 + i * 16) * BK);
        wf1[i][1] = *reinterpret_cast<const bf16x8*>(tS + woff + c1 + (32 + i * 16) * BK);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i][ks], af[j][ks], acc[0][i][j], 0, 0, 0);
      if (!(VAR & 512) && more) stage_w(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[0][2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i][ks], af[j][ks], acc[0][2 + i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        af[j][0] = *reinterpret_cast<const bf16x8*>(tS + aoff + c0 + (64 + j * 16) * BK);
        af[j][1] = *reinterpret_cast<const bf16x8*>(tS + aoff + c1 + (64 + j * 16) * BK);
      }
      if (!(VAR & 512) && more) stage_w(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[1][2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i][ks], af[j][ks], acc[1][2 + i][j], 0, 0, 0);
      if (!(VAR & 512) && more) stage_a1();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i][ks], af[j][ks], acc[1][i][j], 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (last) {
        const int tile_m = c_tile / tiles_n, tile_n = c_tile - tile_m * tiles_n;
        bf16* img = smem + cs * STG + wave * 4096;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m0 = tile_m * BM8 + wp * 128 + h * 64, n0 = tile_n * TN + wc * 64;
          if (bias || (EPI != EPI_NONE && EPI != EPI_BN_TAIL))
            nt_epilogue_lds<MODE, 4, EPI, true>(g, acc[h], m0, n0, 0, 0, lane, bias, out, relu, beta, img, bn_partial,
                                                tail_y, tail_bits);
          else
            nt_epilogue_lds<MODE, 4, EPI, false>(g, acc[h], m0, n0, 0, 0, lane, nullptr, out, relu, beta, img,
                                                 bn_partial, tail_y, tail_bits);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        c_kt = 0;
        c_tile += G;
      } else {
        ++c_kt;
      }
      cs ^= 1;
    }
    if ((VAR & 2048) && lane == 0 && tail_bits) {
      const unsigned long long ck1 = __builtin_readcyclecounter(), rt1 = __builtin_amdgcn_s_memrealtime();
      unsigned long long* d = reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(tail_bits)) + (blockIdx.x * 8 + wave) * 4;
      d[0] = ck1 - ck0;
      d[1] = rt1 - rt0;
    }
    return;
  }

  // ---- prologue: k-tile 0 whole, units A0 and W0 of k-tile 1 ----
  if (cu.tile < ntiles) {
    stage_a0();
    stage_w(0);
    stage_w(1);
    stage_a1();
  }
  if (cu.tile < ntiles) {
    stage_a0();
    stage_w(0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (!(VAR & 4) && wp) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind the first

  int c_tile = vblock, c_kt = 0, cs = 0;
  STAMP(t0);
  bool fresh = true;  // k-tile whose units have all been waited for already (the first one after an epilogue)
  while (c_tile < ntiles) {
    const bool last = c_kt == nk - 1;
    const bf16* tS = smem + cs * STG;

    // ---- phase 1: quadrant (h0, c0) ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      wf0[i][0] = ldfrag<VAR>(tS + woff + c0 + i * 16 * BK);
      wf0[i][1] = ldfrag<VAR>(tS + woff + c1 + i * 16 * BK);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      af[j][0] = ldfrag<VAR>(tS + aoff + c0 + j * 16 * BK);
      af[j][1] = ldfrag<VAR>(tS + aoff + c1 + j * 16 * BK);
    }
    if (!(VAR & 1) && cu.tile < ntiles) stage_w(1);
    STAMP(tw);
    if (!fresh) wait_stream();
    STAMP(tr);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(tm);
    __builtin_amdgcn_sched_barrier(0);
    if (!(VAR & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[0][i][j] = vmfma<VAR>(wf0[i][ks], af[j][ks], acc[0][i][j]);
    if (!(VAR & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(te);
    __builtin_amdgcn_s_barrier();
    PHASE_END();

    // ---- phase 2: quadrant (h0, c1) ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      wf1[i][0] = ldfrag<VAR>(tS + woff + c0 + (32 + i * 16) * BK);
      wf1[i][1] = ldfrag<VAR>(tS + woff + c1 + (32 + i * 16) * BK);
    }
    if (!(VAR & 1) && cu.tile < ntiles) stage_a1();
    if ((VAR & 1) && cu.tile < ntiles) advance(cu);
    STAMP(tw);
    if (!fresh) wait_stream();
    STAMP(tr);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(tm);
    __builtin_amdgcn_sched_barrier(0);
    if (!(VAR & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[0][2 + i][j] = vmfma<VAR>(wf1[i][ks], af[j][ks], acc[0][2 + i][j]);
    if (!(VAR & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(te);
    __builtin_amdgcn_s_barrier();
    PHASE_END();

    // ---- phase 3: quadrant (h1, c1) ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      af[j][0] = ldfrag<VAR>(tS + aoff + c0 + (64 + j * 16) * BK);
      af[j][1] = ldfrag<VAR>(tS + aoff + c1 + (64 + j * 16) * BK);
    }
    if (!(VAR & 1) && !last && cu.tile < ntiles) stage_a0();
    STAMP(tw);
    STAMP(tr);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(tm);
    __builtin_amdgcn_sched_barrier(0);
    if (!(VAR & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[1][2 + i][j] = vmfma<VAR>(wf1[i][ks], af[j][ks], acc[1][2 + i][j]);
    if (!(VAR & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(te);
    __builtin_amdgcn_s_barrier();
    PHASE_END();

    // ---- phase 4: quadrant (h1, c0) ----
    if (last) {
      STAMP(tw);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next k-tile is complete before the epilogue starts
    } else {
      if (!(VAR & 1) && cu.tile < ntiles) stage_w(0);
      STAMP(tw);
      wait_stream();
    }
    STAMP(tr);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    STAMP(tm);
    if (!(VAR & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[1][i][j] = vmfma<VAR>(wf0[i][ks], af[j][ks], acc[1][i][j]);
    if (!(VAR & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(te);
    __builtin_amdgcn_s_barrier();
    PHASE_END();

    fresh = false;
    if (last) {
      // every read of stage cs is two barriers old: the wave parks its output image there (8 KiB per wave)
      const int tile_m = c_tile / tiles_n, tile_n = c_tile - tile_m * tiles_n;
      bf16* img = smem + cs * STG + wave * 4096;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m0 = tile_m * BM8 + wp * 128 + h * 64, n0 = tile_n * TN + wc * 64;
        if (bias || (EPI != EPI_NONE && EPI != EPI_BN_TAIL))
          nt_epilogue_lds<MODE, 4, EPI, true>(g, acc[h], m0, n0, 0, 0, lane, bias, out, relu, beta, img, bn_partial,
                                              tail_y, tail_bits);
        else
          nt_epilogue_lds<MODE, 4, EPI, false>(g, acc[h], m0, n0, 0, 0, lane, nullptr, out, relu, beta, img,
                                               bn_partial, tail_y, tail_bits);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
      if (!(VAR & 1) && cu.tile < ntiles) {  // the two units held back
        stage_a0();
        stage_w(0);
      }
      c_kt = 0;
      c_tile += G;
      fresh = true;
      if (VAR & 64) { unsigned long long tn = __builtin_readcyclecounter(); pa_epi += (unsigned)(tn - t0); t0 = tn; }
    } else {
      ++c_kt;
    }
    cs ^= 1;
  }
  if (!(VAR & 4) && !wp) __builtin_amdgcn_s_barrier();
  if ((VAR & 2048) && lane == 0 && tail_bits) {
    const unsigned long long ck1 = __builtin_readcyclecounter(), rt1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* d = reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(tail_bits)) + (blockIdx.x * 8 + wave) * 4;
    d[0] = ck1 - ck0;
    d[1] = rt1 - rt0;
  }
  if (VAR & 1024) asm volatile("s_waitcnt vmcnt(0)" ::"v"(dummy));
  if ((VAR & 64) && lane == 0 && tail_bits) {
    unsigned* d = reinterpret_cast<unsigned*>(const_cast<uint8_t*>(tail_bits)) + (blockIdx.x * 8 + wave) * 8;
    d[0] = pa_issue; d[1] = pa_wait; d[2] = pa_b1; d[3] = pa_mfma; d[4] = pa_b2; d[5] = pa_n; d[6] = pa_epi; d[7] = 0;
  }
#undef STAMP
#undef PHASE_END
}

// VC_NT8=1 sends the eligible layers to this kernel (read at every call so that one process can run both forms).
// Off by default: measured on the ResNet-50 layer table it is slower than the 128-wide persistent kernel on all but the
// deepest single-tap layers (see the header of this file).
int nt8_enabled() {
  const char* e = getenv("VC_NT8");
  return e ? atoi(e) : 0;
}

}  // namespace

bool vc_nt8_eligible(const ConvGeom& g, int mode, int out_f32) {
  if (!nt8_enabled() || out_f32 || g.ksplit != 1) return false;
  const int taps = g.kd * g.kh * g.kw;
  if (taps > 32 || g.ldi % 8 != 0 || g.K < 8) return false;
  if (!(g.Ci % BK == 0 || (taps == 1 && g.Ci % 8 == 0))) return false;
  if (mode == 1 && (g.sd != 1 || g.sh != 1 || g.sw != 1)) return false;
  const int64_t abytes = (int64_t)g.N * g.Di * g.Hi * g.Wi * g.ldi * 2, wbytes = (int64_t)g.Co * g.ldw * 2;
  if (abytes >= 0xFFFF0000ll || wbytes >= 0xFFFF0000ll) return false;
  return g.Co >= 192;
}

void vc_nt8_launch(const ConvGeom& g, int mode, int epi, hipStream_t stream, const void* src, const void* wgt,
                   const float* bias, void* out, int relu, int beta, float* bn_partial, const void* tail_y,
                   const uint8_t* tail_bits) {
  const int tiles = ceil_div(g.M, BM8) * ceil_div(g.Co, 256);
  const int cus = vc_device_cus();
  const int grid = tiles < cus ? tiles : cus;
  const char* ev = getenv("VC_NT8_VAR");
  const int var = ev ? atoi(ev) : 0;
  if (var == 64 || (var & 2048)) {  // diagnostic builds write their stamps through this pointer
    const char* dbg = getenv("VC_NT8_DBG");
    if (!dbg) return;
    tail_bits = (const uint8_t*)strtoull(dbg, nullptr, 16);
  }
#define VC_NT8_VARCASE(V_)                                                                                        \
  if (mode == 0 && epi == EPI_NONE && var == V_) {                                                                \
    hipLaunchKernelGGL((igemm_nt8_kernel<0, EPI_NONE, V_>), dim3(grid), dim3(NT8), 0, stream, g, (const bf16*)src, \
                       (const bf16*)wgt, bias, (bf16*)out, relu, beta, tiles, bn_partial, (const bf16*)tail_y,    \
                       tail_bits);                                                                                \
    return;                                                                                                       \
  }
#ifdef VC_NT8_DIAG  // ablation / stamp builds of the forward kernel (make CXXFLAGS+=-DVC_NT8_DIAG)
  VC_NT8_VARCASE(1) VC_NT8_VARCASE(2) VC_NT8_VARCASE(3) VC_NT8_VARCASE(4) VC_NT8_VARCASE(8) VC_NT8_VARCASE(16)
  VC_NT8_VARCASE(18) VC_NT8_VARCASE(48) VC_NT8_VARCASE(64) VC_NT8_VARCASE(768) VC_NT8_VARCASE(1024) VC_NT8_VARCASE(2048)
  VC_NT8_VARCASE(2049) VC_NT8_VARCASE(2050) VC_NT8_VARCASE(2051) VC_NT8_VARCASE(2064) VC_NT8_VARCASE(2304)
#endif
#undef VC_NT8_VARCASE
#define VC_NT8_CASE(M_, E_)                                                                                       \
  if (mode == M_ && epi == E_) {                                                                                  \
    if (var == 256)                                                                                               \
      hipLaunchKernelGGL((igemm_nt8_kernel<M_, E_, 256>), dim3(grid), dim3(NT8), 0, stream, g, (const bf16*)src,  \
                         (const bf16*)wgt, bias, (bf16*)out, relu, beta, tiles, bn_partial, (const bf16*)tail_y,  \
                         tail_bits);                                                                              \
    else                                                                                                          \
      hipLaunchKernelGGL((igemm_nt8_kernel<M_, E_, 0>), dim3(grid), dim3(NT8), 0, stream, g, (const bf16*)src,    \
                         (const bf16*)wgt, bias, (bf16*)out, relu, beta, tiles, bn_partial, (const bf16*)tail_y,  \
                         tail_bits);                                                                              \
    return;                                                                                                       \
  }
  VC_NT8_CASE(0, EPI_NONE)
  VC_NT8_CASE(1, EPI_NONE)
#undef VC_NT8_CASE
}
