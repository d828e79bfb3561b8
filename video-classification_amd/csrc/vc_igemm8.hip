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
  int fake_pos = wave * 1024;
  auto DMA = [&](const __amdgpu_buffer_rsrc_t& rs, bf16* lds, int voff, int soff) {
    if ((VAR & 4096) || ((VAR & 8192) && &rs == &rs_a) || ((VAR & 16384) && &rs == &rs_w)) {  // timing experiment: the same requests against a 1 MiB window at the start of the operand
      fake_pos = (fake_pos + 8192) & 0xFFFFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, (int)((lane >> 3) * 512 + ((VAR & 32768) ? lchunk : (lane & 7)) * 16), fake_pos, 0, 0);
    } else if (VAR & 1024) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dummy) : "v"(voff), "s"(rs), "s"(soff));
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

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined form of the same tile (VC_NT8_VAR=512): no wave row is held behind the other, ONE barrier per
// k-tile.  A k-tile is two 32-deep halves; while a wave multiplies one half out of one fragment set (12 registers
// quads: 8 pixel tiles + 4 channel tiles) it reads the other half -- or the first half of the NEXT k-tile -- into the
// second set, one read after every other product, and issues the 8 requests of the k-tile after next one at a time
// between products of the second half.  The barrier sits between the halves: by then every wave has read both halves
// of the current k-tile (its stage may be overwritten) and waited for its own requests of the next one.
// The epilogue has an LDS region of its own (4 KiB per wave), so the request stream never pauses.
// ---------------------------------------------------------------------------------------------------------------
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

template <int MODE, int EPI, int VAR = 0>
__global__ __launch_bounds__(NT8, 2) void igemm_nt8p_kernel(ConvGeom g, const bf16* __restrict__ src,
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
    if (VAR & 2) {  // timing only: no requests
      if (p == 7 && live) advance(cu);
      return;
    }
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

  // VAR & 16: the 64 requests of a k-tile (request r = piece * 8 + wave) go out one wave at a time, two every three
  // products: slot (2 r) / 3 of the 43 that follow the barrier, the last 11 of them in the next k-tile's first half
  bool pend = false;
  auto issue_slot = [&](int slot, bool live) __attribute__((always_inline)) {
    const int r0 = (3 * slot + 1) / 2, r1 = r0 + 1;  // the one or two requests of this slot
    if ((2 * r0) / 3 == slot && r0 < 64 && wave == (r0 & 7)) stage_piece(r0 >> 3, live);
    if ((2 * r1) / 3 == slot && r1 < 64 && wave == (r1 & 7)) stage_piece(r1 >> 3, live);
  };
  while (c_tile < ntiles) {
    const bool last = c_kt == nk - 1;
    const bf16* tS = smem + cs * STG;
    const bf16* tN = smem + (cs ^ 1) * STG;
    // ---- first half: products out of X, second half's fragments into Y ----
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int ci = n & 3, jj = n >> 2;
      acc[jj >> 2][ci][jj & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfX[ci], afX[jj], acc[jj >> 2][ci][jj & 3], 0, 0, 0);
      if ((n & 1) == 0 && n < 24 && !(VAR & 4)) ldY(tS, n >> 1, c1);
      if ((VAR & 16) && n < 11) issue_slot(32 + n, pend);
      __builtin_amdgcn_sched_barrier(0);
    }
    // every read of this k-tile's stage is complete, and so are this wave's requests of the next k-tile
    if (VAR & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // timing only: no barrier
    else if (VAR & 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // timing only: requests not waited for
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // is there a k-tile after this one (for this workgroup)?  its first half goes into X
    const bool more = cu.tile < ntiles;
    if (more) stage_prepare();
    pend = more;
    // ---- second half: products out of Y, the next k-tile's first fragments into X, requests of the one after ----
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int ci = n & 3, jj = n >> 2;
      acc[jj >> 2][ci][jj & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfY[ci], afY[jj], acc[jj >> 2][ci][jj & 3], 0, 0, 0);
      if ((n & 1) == 0 && n < 24 && !(VAR & 4)) ldX(tN, n >> 1, c0);  // past the last k-tile: stale LDS, never multiplied
      if (VAR & 16) issue_slot(n, more);
      else if ((n & 3) == 1) stage_piece(n >> 2, more);
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

// ---------------------------------------------------------------------------------------------------------------
// Row-unit form (VC_NT8_VAR=768): the same two 64 KiB stages, but a stage is given back in three parts as soon as the
// last wave has read them, so that every request has a whole k-tile to land in and requests are in flight all the time.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, int EPI, int VAR = 0>
__global__ __launch_bounds__(NT8, 2) void igemm_nt8r_kernel(ConvGeom g, const bf16* __restrict__ src,
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
  // Lane-dependent addressing terms are re-derived from the thread index at the start of every block (the index passes
  // through an empty asm, so that the compiler cannot keep the derived values in registers across the whole loop, where
  // they would be spilled and each reload would wait for every outstanding request):
  int lrow = 0, lc16 = 0, lpart = 0;  // staging row of 64, 16 x the logical chunk it fetches, fragment address part
  auto derive = [&]() __attribute__((always_inline)) {
    int t = tid;
    asm volatile("" : "+v"(t));
    lrow = t >> 3;
    lc16 = ((t & 7) ^ ((t >> 4) & 7)) << 4;
    const int f15 = t & 15, fq = (t >> 4) & 3;
    lpart = (f15 * BK + ((fq ^ ((f15 >> 1) & 7)) << 3)) * 2;
  };
  derive();
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
  unsigned wvo0 = 0;
  uint2 rowent = make_uint2(0u, 0u);  // table entry of the next pixel-row request, read one product ahead of it
  int s_kb = 0, s_wk = 0, s_toff = 0;
  bool s_dead = false;
  const i32x4 rs_a = {(int)(uintptr_t)src, (int)((uintptr_t)src >> 32),
                      (int)((unsigned)(g.N * g.Di * g.Hi * g.Wi) * (unsigned)g.ldi * 2u), 0x00020000};
  const i32x4 rs_w = {(int)(uintptr_t)wgt, (int)((uintptr_t)wgt >> 32), (int)((unsigned)g.Co * (unsigned)g.ldw * 2u),
                      0x00020000};

  // state of the cursor's k-tile (rows of a new output tile, offsets of a new tap): before its first request
  // rows of output tile `tile` into table `par` (one row per thread of the first four waves)
  auto write_rows = [&](int tile, int par) {
    int t = tid;
    asm volatile("" : "+v"(t));
    if (t < BM8 && tile < ntiles) {
      const RowPos r = decode_row<MODE>(g, (tile / tiles_n) * BM8 + t);
      rowtab[par][t] = make_uint2((unsigned)((((r.n * g.Di + r.d0) * g.Hi + r.h0) * g.Wi + r.w0) * g.ldi * 2),
                                    tap_mask<MODE>(g, r));
    }
  };
  // state of the cursor's k-tile (offsets of a new tap, channel row of a new output tile): before its first request.
  // The rows of the tile after the cursor's are decoded while its last k-tile is requested: a barrier lies between
  // that and the first use.
  auto stage_prepare = [&]() {
    if (cu.kt == 0) {
      const int n0 = (cu.tile % tiles_n) * TN;
      wvo0 = (unsigned)((n0 + (lrow >> 5) * 64 + (lrow & 31)) * g.ldw * 2 + lc16);
    }
    if (cu.cb == 0) {
      s_toff = ((cu.rd * g.Hi + cu.rh) * g.Wi + cu.rw) * g.ldi * 2;
      if (MODE != 0) s_toff = -s_toff;
      s_wk = MODE == 2 ? ((cu.rd * g.wt_kh + g.wt_h0 + g.wt_hs * cu.rh) * g.wt_kw + g.wt_w0 + g.wt_ws * cu.rw) * g.Ci * 2
                       : cu.tap * g.Ci * 2;
    }
    s_kb = cu.cb * (BK * 2);
    s_dead = ktail && cu.cb * BK + (lc16 >> 1) >= g.Ci;
    if (cu.kt == nk - 1) write_rows(cu.tile + G, cu.par ^ 1);
  };
  // request p (0..7) of the cursor's k-tile: 0,1 pixel rows [0,64) of both wave rows; 2,3 / 4,5 the first / second 32
  // channels of every wave column; 6,7 pixel rows [64,128); the last one moves the cursor on
  // the table entry request p (0, 1, 6, 7: pixel rows) will need; issued one product before the request
  auto stage_row = [&](int p) __attribute__((always_inline)) {
    rowent = rowtab[cu.par][(p & 1) * 128 + (p >= 6 ? 64 : 0) + lrow];
  };
  auto stage_piece = [&](int p, bool live) __attribute__((always_inline)) {
    bf16* sA = smem + cu.st * STG;
    bf16* sW = sA + ASZ;
    const int gg = p & 1;
    if (p < 2 || p >= 6) {
      const bool ok = ((rowent.y >> cu.tap) & 1u) && !s_dead && live;
      dma16(rs_a, lds_addr(sA + (gg * 128 + (p >= 6 ? 64 : 0) + 8 * wave) * BK),
            ok ? rowent.x + (unsigned)(s_toff + lc16) : SENT, s_kb);
      if (p == 7 && live) advance(cu);
    } else {
      const int c = (p - 2) >> 1;
      dma16(rs_w, lds_addr(sW + ((gg * 2 + (wave >> 2)) * 64 + c * 32 + (wave & 3) * 8) * BK),
            s_dead || !live ? SENT : wvo0 + (unsigned)((gg * 128 + c * 32) * g.ldw * 2), s_wk + s_kb);
    }
  };

  f32x4 acc[2][4][4];  // [pixel half][channel tile][pixel tile]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};


  // Fragments: the wave's 4 channel tiles for both halves of k (wf, replaced tile by tile as the second block of a
  // k-tile finishes with them), and two sets of 4 pixel tiles x 2 halves of k (af0: pixel rows [0,64), af1: [64,128)).
  bf16x8 wf[2][4], af0[2][4], af1[2][4];
  // One register holds the lane's part of every fragment address (row l15, chunk of the first k half); the wave's row
  // base, the stage and the second k half (chunk ^ 4, i.e. byte address ^ 64) are applied where a block reads, so that
  // no second address register stays live across the blocks.
  auto frag = [&](unsigned byte) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(smem) + byte);
  };
  auto ld_w = [&](int st, int k, int ci) __attribute__((always_inline)) {
    const unsigned b0 = (unsigned)lpart + (unsigned)((st * STG + ASZ + wc * 64 * BK) * 2);
    wf[k][ci] = frag((k ? b0 ^ 64u : b0) + ci * 16 * BK * 2);
  };
  auto ld_a0 = [&](int st, int k, int j) __attribute__((always_inline)) {
    const unsigned b0 = (unsigned)lpart + (unsigned)((st * STG + wp * 128 * BK) * 2);
    af0[k][j] = frag((k ? b0 ^ 64u : b0) + j * 16 * BK * 2);
  };
  auto ld_a1 = [&](int st, int k, int j) __attribute__((always_inline)) {
    const unsigned b0 = (unsigned)lpart + (unsigned)((st * STG + wp * 128 * BK) * 2);
    af1[k][j] = frag((k ? b0 ^ 64u : b0) + (64 + j * 16) * BK * 2);
  };

  // ---- prologue: k-tiles 0 and 1 requested, k-tile 0 waited for, its channel tiles and first pixel half in registers
  write_rows(cu.tile, 0);
  __syncthreads();
  {
    const bool more = cu.tile < ntiles;
    if (more) stage_prepare();
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      if (p < 2 || p >= 6) stage_row(p);
      stage_piece(p, more);
    }
  }
  __syncthreads();
  {
    const bool more = cu.tile < ntiles;
    if (more) stage_prepare();
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      if (p < 2 || p >= 6) stage_row(p);
      stage_piece(p, more);
    }
  }
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  int c_tile = vblock, c_kt = 0, cs = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    ld_w(0, r >> 2, r & 3);
    ld_a0(0, r >> 2, r & 3);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // Per k-tile t (stage s) two blocks of 32 products, a barrier after each:
  //  block A: pixel rows [0,64) out of (wf, af0); reads af1 of t; requests [0,64)-rows and the first 32 channels of every
  //           wave column of k-tile t + 2 into stage s (their last reads were in block B of t - 1);
  //  block B: pixel rows [64,128) out of (wf, af1); reads af0 and wf of t + 1 (stage s ^ 1); requests the other 32
  //           channels and rows [64,128) of t + 2 (rows [64,128) of t were last read in block A).
  // Waits: end of A allows the 2 row requests of the previous B and A's own 4 (vmcnt 6): everything block B reads has
  // landed; end of B allows A's and its own (vmcnt 8): rows [64,128) of the next k-tile have landed.
  while (c_tile < ntiles) {
    const bool more = cu.tile < ntiles;
    derive();
    if (more) stage_prepare();
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int ci = n >> 3, k = (n >> 2) & 1, j = n & 3;
      acc[0][ci][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k][ci], af0[k][j], acc[0][ci][j], 0, 0, 0);
      if ((n & 1) == 0 && n < 16) ld_a1(cs, n >> 3, (n >> 1) & 3);
      if (n == 0 || n == 4) stage_row(n >> 2);
      if ((n & 3) == 1 && n < 16) stage_piece(n >> 2, more);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    derive();
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int ci = n >> 3, k = (n >> 2) & 1, j = n & 3;
      acc[1][ci][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k][ci], af1[k][j], acc[1][ci][j], 0, 0, 0);
      if ((n & 1) == 0 && n < 16) ld_a0(cs ^ 1, n >> 3, (n >> 1) & 3);  // past the last k-tile: stale LDS, never multiplied
      if (n >= 8 && (n & 7) < 4 && (n & 1) == 1) ld_w(cs ^ 1, (n >> 1) & 1, (n >> 3) - 1);  // tile ci - 1 is done with
      if (n == 12 || n == 20) stage_row(n == 12 ? 6 : 7);
      if (n == 1 || n == 5 || n == 13 || n == 21) stage_piece(n == 1 ? 4 : n == 5 ? 5 : n == 13 ? 6 : 7, more);
      __builtin_amdgcn_sched_barrier(0);
    }
    ld_w(cs ^ 1, 0, 3);
    ld_w(cs ^ 1, 1, 3);
    if (more) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (++c_kt == nk) {
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
    }
    cs ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// Ring form (VC_NT8_VAR=1024): the k dimension is staged in slabs of 32, four slabs deep (4 x 32 KiB), rows of 64 bytes
// with the 16-byte chunk of row r at position chunk ^ f((r >> 2) & 3), f = {0, 2, 3, 1}: conflict-free for the lane groups
// a 16-row ds_read_b128 is served in ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, and the same plus 32).
// Block u of a wave: 32 products out of slab u's fragments (in registers), the 12 fragment reads of slab u + 1 into the
// other register set, the wave's 4 requests of slab u + 4 into the slot slab u has left; then lgkmcnt(0), vmcnt(8) --
// slab u + 2 has landed, u + 3 and u + 4 stay in flight -- and ONE barrier.  A request has two blocks to land in.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, int EPI, int VAR = 0>
__global__ __launch_bounds__(NT8, 2) void igemm_nt8q_kernel(ConvGeom g, const bf16* __restrict__ src,
                                                            const bf16* __restrict__ wgt, const float* __restrict__ bias,
                                                            bf16* __restrict__ out, int relu, int beta, int ntiles,
                                                            float* __restrict__ bn_partial,
                                                            const bf16* __restrict__ tail_y,
                                                            const uint8_t* __restrict__ tail_bits) {
  constexpr int TN = 256, SK = 32;
  constexpr int ASZ = BM8 * SK, WSZ = TN * SK, SLAB = ASZ + WSZ;  // elements per slab
  constexpr int EPW = 1024;                                        // epilogue image per wave: 64 pixels x 16 channels
  __shared__ __attribute__((aligned(1024))) bf16 smem[4 * SLAB + 8 * EPW];
  __shared__ uint2 rowtab[2][BM8];  // per row of an output tile: byte offset of its window origin, tap mask

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 2, wc = wave & 3;
  const int tiles_n = (g.Co + TN - 1) / TN;
  int vblock;
  xcd_remap(vblock);
  const int G = gridDim.x;
  const int r16 = lane >> 2;                            // row of a request's 16
  auto swz = [](int r2) { return (((r2 >> 1) ^ r2) & 1) << 1 | (r2 >> 1); };  // {0, 2, 3, 1}
  const int lchunk = (lane & 3) ^ swz((lane >> 4) & 3);  // logical 16-byte chunk this lane fetches
  const int ncb = 2 * ((g.Ci + 63) / 64);               // slabs per tap (even: a tile ends on an odd block)
  const int ntaps = g.kd * g.kh * g.kw;
  const int nk = ntaps * ncb;
  const bool ktail = (g.Ci % 64) != 0;

  struct KCursor {
    int tile, kt, tap, cb, rd, rh, rw, slot, par;
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
      c.par ^= 1;
    }
    c.slot = (c.slot + 1) & 3;
  };
  KCursor cu{vblock, 0, 0, 0, 0, 0, 0, 0, 0};

  constexpr unsigned SENT = 0xFFFFFFF0u;
  unsigned avo[2], wvo0 = 0;
  int s_kb = 0, s_wk = 0;
  bool s_dead = false;
  const i32x4 rs_a = {(int)(uintptr_t)src, (int)((uintptr_t)src >> 32),
                      (int)((unsigned)(g.N * g.Di * g.Hi * g.Wi) * (unsigned)g.ldi * 2u), 0x00020000};
  const i32x4 rs_w = {(int)(uintptr_t)wgt, (int)((uintptr_t)wgt >> 32), (int)((unsigned)g.Co * (unsigned)g.ldw * 2u),
                      0x00020000};

  auto write_rows = [&](int tile, int par) {
    if (tid < BM8 && tile < ntiles) {
      const RowPos r = decode_row<MODE>(g, (tile / tiles_n) * BM8 + tid);
      rowtab[par][tid] = make_uint2((unsigned)((((r.n * g.Di + r.d0) * g.Hi + r.h0) * g.Wi + r.w0) * g.ldi * 2),
                                    tap_mask<MODE>(g, r));
    }
  };
  // state of the cursor's slab, before its first request.  The rows of the tile after the cursor's are decoded while
  // its last slab is requested: a barrier lies between that and their first use.
  auto stage_prepare = [&]() {
    if (cu.kt == 0) {
      const int n0 = (cu.tile % tiles_n) * TN;
      wvo0 = (unsigned)((n0 + wave * 16 + r16) * g.ldw * 2 + lchunk * 16);
    }
    if (cu.cb == 0) {
      int toff = ((cu.rd * g.Hi + cu.rh) * g.Wi + cu.rw) * g.ldi * 2;
      if (MODE != 0) toff = -toff;
      toff += lchunk * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint2 e = rowtab[cu.par][i * 128 + wave * 16 + r16];
        avo[i] = ((e.y >> cu.tap) & 1u) ? e.x + (unsigned)toff : SENT;
      }
      s_wk = MODE == 2 ? ((cu.rd * g.wt_kh + g.wt_h0 + g.wt_hs * cu.rh) * g.wt_kw + g.wt_w0 + g.wt_ws * cu.rw) * g.Ci * 2
                       : cu.tap * g.Ci * 2;
    }
    s_kb = cu.cb * (SK * 2);
    s_dead = ktail && cu.cb * SK + lchunk * 8 >= g.Ci;
    if (cu.kt == nk - 1) write_rows(cu.tile + G, cu.par ^ 1);
  };
  // request p (0..3) of the cursor's slab: 0,1 pixel rows [0,128) / [128,256), 2,3 channels likewise; a request moves
  // 16 rows per wave.  Requests of a slab past the last tile fetch nothing (zeros land in a slot nobody reads), so that
  // every block issues the same number.
  auto stage_piece = [&](int p, bool live) {
    bf16* sA = smem + cu.slot * SLAB;
    const int gg = p & 1;
    if (p < 2)
      dma16(rs_a, lds_addr(sA + (gg * 128 + wave * 16) * SK), s_dead || !live ? SENT : avo[gg], s_kb);
    else {
      dma16(rs_w, lds_addr(sA + ASZ + (gg * 128 + wave * 16) * SK),
            s_dead || !live ? SENT : wvo0 + (unsigned)(gg * 128 * g.ldw * 2), s_wk + s_kb);
      if (p == 3 && live) advance(cu);
    }
  };

  f32x4 acc[2][4][4];  // [pixel half][channel tile][pixel tile]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, q = lane >> 4;
  const int pc = (q ^ swz((l15 >> 2) & 3)) << 3;
  const int aoff = (wp * 128 + l15) * SK + pc, woff = ASZ + (wc * 64 + l15) * SK + pc;

  bf16x8 afX[8], wfX[4], afY[8], wfY[4];
  // fragment r of a slab: 0..3 the wave's channel tiles, 4..11 its pixel tiles
  auto ldf = [&](bf16x8(&wf)[4], bf16x8(&af)[8], const bf16* tS, int r) __attribute__((always_inline)) {
    if (r < 4) wf[r] = *reinterpret_cast<const bf16x8*>(tS + woff + r * 16 * SK);
    else af[r - 4] = *reinterpret_cast<const bf16x8*>(tS + aoff + (r - 4) * 16 * SK);
  };

  // ---- prologue: slabs 0..3 requested, 0 and 1 waited for, slab 0's fragments in registers ----
  write_rows(cu.tile, 0);
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    __syncthreads();
    const bool more = cu.tile < ntiles;
    if (more) stage_prepare();
#pragma unroll
    for (int p = 0; p < 4; ++p) stage_piece(p, more);
  }
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  int c_tile = vblock, c_kt = 0, cs = 0;
#pragma unroll
  for (int r = 0; r < 12; ++r) ldf(wfX, afX, smem, r);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // one block: products out of (wf, af), the next slab's fragments into (nwf, naf), requests into the current slot
  auto block = [&](bf16x8(&wf)[4], bf16x8(&af)[8], bf16x8(&nwf)[4], bf16x8(&naf)[8]) __attribute__((always_inline)) {
    const bf16* tN = smem + ((cs + 1) & 3) * SLAB;
    const bool more = cu.tile < ntiles;
    if (more) stage_prepare();
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      const int ci = n & 3, jj = n >> 2;
      acc[jj >> 2][ci][jj & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ci], af[jj], acc[jj >> 2][ci][jj & 3], 0, 0, 0);
      if ((n & 1) == 0 && n < 24) ldf(nwf, naf, tN, n >> 1);  // past the last slab: stale LDS, never multiplied
      if ((n & 3) == 1 && n < 16) stage_piece(n >> 2, more);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (VAR & 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // timing only
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    cs = (cs + 1) & 3;
  };

  while (c_tile < ntiles) {
    block(wfX, afX, wfY, afY);
    block(wfY, afY, wfX, afX);
    c_kt += 2;
    if (c_kt == nk) {
      const int tile_m = c_tile / tiles_n, tile_n = c_tile - tile_m * tiles_n;
      bf16* img = smem + 4 * SLAB + wave * EPW;
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
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
#define VC_NT8P_VARCASE(V_)                                                                                       \
  if (mode == 0 && epi == EPI_NONE && var == 512 + V_) {                                                          \
    hipLaunchKernelGGL((igemm_nt8p_kernel<0, EPI_NONE, V_>), dim3(grid), dim3(NT8), 0, stream, g, (const bf16*)src, \
                       (const bf16*)wgt, bias, (bf16*)out, relu, beta, tiles, bn_partial, (const bf16*)tail_y,    \
                       tail_bits);                                                                                \
    return;                                                                                                       \
  }
  VC_NT8P_VARCASE(16) VC_NT8P_VARCASE(17)
#undef VC_NT8P_VARCASE
#if 0
  VC_NT8_VARCASE(1) VC_NT8_VARCASE(2) VC_NT8_VARCASE(3) VC_NT8_VARCASE(4) VC_NT8_VARCASE(8) VC_NT8_VARCASE(16)
  VC_NT8_VARCASE(18) VC_NT8_VARCASE(48) VC_NT8_VARCASE(64) VC_NT8_VARCASE(768) VC_NT8_VARCASE(1024) VC_NT8_VARCASE(2048)
  VC_NT8_VARCASE(4096) VC_NT8_VARCASE(4114) VC_NT8_VARCASE(36882) VC_NT8_VARCASE(8210) VC_NT8_VARCASE(16402) VC_NT8_VARCASE(2049) VC_NT8_VARCASE(2050) VC_NT8_VARCASE(2051) VC_NT8_VARCASE(2064) VC_NT8_VARCASE(2304)
#endif
#endif
#undef VC_NT8_VARCASE
#define VC_NT8_CASE(M_, E_)                                                                                       \
  if (mode == M_ && epi == E_) {                                                                                  \
    if (var == 768) {                                                                                             \
      hipLaunchKernelGGL((igemm_nt8r_kernel<M_, E_, 0>), dim3(grid), dim3(NT8), 0, stream, g, (const bf16*)src,   \
                         (const bf16*)wgt, bias, (bf16*)out, relu, beta, tiles, bn_partial, (const bf16*)tail_y,  \
                         tail_bits);                                                                              \
    } else if (var == 1024 || var == 1025) {                                                                      \
      if (var == 1024)                                                                                            \
        hipLaunchKernelGGL((igemm_nt8q_kernel<M_, E_, 0>), dim3(grid), dim3(NT8), 0, stream, g, (const bf16*)src, \
                           (const bf16*)wgt, bias, (bf16*)out, relu, beta, tiles, bn_partial, (const bf16*)tail_y,\
                           tail_bits);                                                                            \
      else                                                                                                        \
        hipLaunchKernelGGL((igemm_nt8q_kernel<M_, E_, 1>), dim3(grid), dim3(NT8), 0, stream, g, (const bf16*)src, \
                           (const bf16*)wgt, bias, (bf16*)out, relu, beta, tiles, bn_partial, (const bf16*)tail_y,\
                           tail_bits);                                                                            \
    } else if (var == 513)                                                                                        \
      hipLaunchKernelGGL((igemm_nt8p_kernel<M_, E_, 1>), dim3(grid), dim3(NT8), 0, stream, g, (const bf16*)src,   \
                         (const bf16*)wgt, bias, (bf16*)out, relu, beta, tiles, bn_partial, (const bf16*)tail_y,  \
                         tail_bits);                                                                              \
    else if (var == 512)                                                                                          \
      hipLaunchKernelGGL((igemm_nt8p_kernel<M_, E_, 0>), dim3(grid), dim3(NT8), 0, stream, g, (const bf16*)src,   \
                         (const bf16*)wgt, bias, (bf16*)out, relu, beta, tiles, bn_partial, (const bf16*)tail_y,  \
                         tail_bits);                                                                              \
    else if (var == 256)                                                                                          \
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
