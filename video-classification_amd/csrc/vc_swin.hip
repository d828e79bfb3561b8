// Shifted-window attention and patch merging for the Swin-T backbone (gfx950).
//
// Window partition, the cyclic shift and its inverse are never materialised: the kernels address the
// token matrix [B][H][W][*] through the window/shift index map.  A (window, head) unit is 49 x 49 x 32
// (window 7, head_dim 32): far too small to be worth an MFMA pipeline next to the QKV / projection / MLP
// GEMMs (the attention products are ~3 % of Swin-T's FLOPs), so one wave owns one unit, K/V/Q live in LDS,
// each lane owns one query row and the softmax never leaves registers.
//
// qkv layout (timm): row = token, column = which*C + head*32 + e, which in {q, k, v}.
#include "vc_common.h"

namespace {

constexpr int HD = 32;      // head dim
constexpr int MAXN = 49;    // tokens per window (window 7)
constexpr int LDP = 50;     // padded row length of the [i][j] LDS matrices

struct SwinGeom {
  int B, H, W, C, heads, ws, shift;
  int nwh, nww;  // windows per column / row
};

// token index (row of the [B*H*W] matrices) of local position n in window (b, wh, ww)
__device__ __forceinline__ int token_of(const SwinGeom& g, int b, int wh, int ww, int n, int& ph, int& pw) {
  const int i = n / g.ws, j = n - i * g.ws;
  ph = wh * g.ws + i;  // coordinates in the rolled frame
  pw = ww * g.ws + j;
  int h = ph + g.shift, w = pw + g.shift;
  if (h >= g.H) h -= g.H;
  if (w >= g.W) w -= g.W;
  return (b * g.H + h) * g.W + w;
}

__device__ __forceinline__ int region_of(int p, int L, int ws, int shift) {
  return p < L - ws ? 0 : (p < L - shift ? 1 : 2);
}

template <typename T>
__device__ __forceinline__ void load_row32(const T* p, float* dst) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const F8 v = load8(p + 8 * k);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[8 * k + e] = v.v[e];
  }
}

// scores of query row `lane` against all keys: p[j] = softmax_j(scale*q.k_j + bias + mask); returns nothing,
// p[] holds probabilities on exit.  sK is [N][HD] fp32 in LDS.
__device__ __forceinline__ void scores_softmax(const SwinGeom& g, const float* q, const float* sK,
                                               const float* __restrict__ bias_table, int head, int lane, int N,
                                               int my_reg, const int* sReg, float scale, float* p) {
  const int ws = g.ws;
  const int qi = lane / ws, qj = lane - qi * ws;
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXN; ++j) {
    float s = 0.f;
    if (j < N) {
      const float* kr = sK + j * HD;
#pragma unroll
      for (int e = 0; e < HD; e += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + e);
        s = fmaf(q[e], kv[0], s);
        s = fmaf(q[e + 1], kv[1], s);
        s = fmaf(q[e + 2], kv[2], s);
        s = fmaf(q[e + 3], kv[3], s);
      }
      const int kj_i = j / ws, kj_j = j - kj_i * ws;
      const int idx = (qi - kj_i + ws - 1) * (2 * ws - 1) + (qj - kj_j + ws - 1);
      s = s * scale + bias_table[idx * g.heads + head];
      if (g.shift && sReg[j] != my_reg) s -= 100.f;
      mx = fmaxf(mx, s);
    } else {
      s = -INFINITY;
    }
    p[j] = s;
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < MAXN; ++j) {
    const float e = (j < N) ? __expf(p[j] - mx) : 0.f;
    p[j] = e;
    sum += e;
  }
  const float inv = 1.f / sum;
#pragma unroll
  for (int j = 0; j < MAXN; ++j) p[j] *= inv;
}

// one wave per (window, head)
template <typename T>
__global__ __launch_bounds__(64) void window_attn_fwd_kernel(SwinGeom g, const T* __restrict__ qkv,
                                                             const float* __restrict__ bias_table, T* __restrict__ out,
                                                             float scale) {
  __shared__ __attribute__((aligned(16))) float sK[MAXN * HD];
  __shared__ __attribute__((aligned(16))) float sV[MAXN * HD];
  __shared__ int sReg[64];
  const int lane = threadIdx.x;
  const int N = g.ws * g.ws;
  int unit = blockIdx.x;
  const int head = unit % g.heads;
  unit /= g.heads;
  const int ww = unit % g.nww;
  unit /= g.nww;
  const int wh = unit % g.nwh;
  const int b = unit / g.nwh;

  float q[HD];
  int tok = 0, my_reg = 0;
  if (lane < N) {
    int ph, pw;
    tok = token_of(g, b, wh, ww, lane, ph, pw);
    my_reg = region_of(ph, g.H, g.ws, g.shift) * 3 + region_of(pw, g.W, g.ws, g.shift);
    const T* row = qkv + (size_t)tok * 3 * g.C + head * HD;
    load_row32(row, q);
    load_row32(row + g.C, sK + lane * HD);
    load_row32(row + 2 * g.C, sV + lane * HD);
  } else {
#pragma unroll
    for (int e = 0; e < HD; ++e) q[e] = 0.f;
  }
  sReg[lane] = my_reg;
  __syncthreads();

  float p[MAXN];
  scores_softmax(g, q, sK, bias_table, head, lane < N ? lane : 0, N, my_reg, sReg, scale, p);
  float o[HD];
#pragma unroll
  for (int e = 0; e < HD; ++e) o[e] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXN; ++j) {
    if (j < N) {
      const float* vr = sV + j * HD;
#pragma unroll
      for (int e = 0; e < HD; e += 4) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + e);
        o[e] = fmaf(p[j], vv[0], o[e]);
        o[e + 1] = fmaf(p[j], vv[1], o[e + 1]);
        o[e + 2] = fmaf(p[j], vv[2], o[e + 2]);
        o[e + 3] = fmaf(p[j], vv[3], o[e + 3]);
      }
    }
  }
  if (lane < N) {
    T* orow = out + (size_t)tok * g.C + head * HD;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      F8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v.v[e] = o[8 * k + e];
      store8(orow + 8 * k, v);
    }
  }
}

// One wave per (head, chunk of windows): recomputes the probabilities, produces dqkv for its windows and
// accumulates the relative-position-bias gradient of its windows in registers (lane i owns row i of dS).
template <typename T>
__global__ __launch_bounds__(64) void window_attn_bwd_kernel(SwinGeom g, const T* __restrict__ qkv,
                                                             const T* __restrict__ dout,
                                                             const float* __restrict__ bias_table,
                                                             T* __restrict__ dqkv, float* __restrict__ dbias_table,
                                                             float scale, int windows_total, int windows_per_block) {
  __shared__ __attribute__((aligned(16))) float sQ[MAXN * HD];
  __shared__ __attribute__((aligned(16))) float sK[MAXN * HD];
  __shared__ __attribute__((aligned(16))) float sV[MAXN * HD];
  __shared__ __attribute__((aligned(16))) float sDO[MAXN * HD];
  __shared__ float sP[MAXN * LDP];
  __shared__ float sDS[MAXN * LDP];
  __shared__ int sReg[64];
  const int lane = threadIdx.x;
  const int N = g.ws * g.ws;
  const int head = blockIdx.y;
  const int w_begin = blockIdx.x * windows_per_block;
  const int w_end = min(windows_total, w_begin + windows_per_block);

  float db[MAXN];
#pragma unroll
  for (int j = 0; j < MAXN; ++j) db[j] = 0.f;

  for (int win = w_begin; win < w_end; ++win) {
    int u = win;
    const int ww = u % g.nww;
    u /= g.nww;
    const int wh = u % g.nwh;
    const int b = u / g.nwh;
    float q[HD], dO[HD];
    int tok = 0, my_reg = 0;
    __syncthreads();  // previous window's LDS reads are finished
    if (lane < N) {
      int ph, pw;
      tok = token_of(g, b, wh, ww, lane, ph, pw);
      my_reg = region_of(ph, g.H, g.ws, g.shift) * 3 + region_of(pw, g.W, g.ws, g.shift);
      const T* row = qkv + (size_t)tok * 3 * g.C + head * HD;
      load_row32(row, q);
      load_row32(row + g.C, sK + lane * HD);
      load_row32(row + 2 * g.C, sV + lane * HD);
      load_row32(dout + (size_t)tok * g.C + head * HD, dO);
#pragma unroll
      for (int e = 0; e < HD; ++e) {
        sQ[lane * HD + e] = q[e];
        sDO[lane * HD + e] = dO[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < HD; ++e) q[e] = dO[e] = 0.f;
    }
    sReg[lane] = my_reg;
    __syncthreads();

    float p[MAXN];
    scores_softmax(g, q, sK, bias_table, head, lane < N ? lane : 0, N, my_reg, sReg, scale, p);
    // dP[i][j] = dO_i . V_j ;  dS = P * (dP - sum_j P dP)
    float ds[MAXN];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      float d = 0.f;
      if (j < N) {
        const float* vr = sV + j * HD;
#pragma unroll
        for (int e = 0; e < HD; e += 4) {
          const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + e);
          d = fmaf(dO[e], vv[0], d);
          d = fmaf(dO[e + 1], vv[1], d);
          d = fmaf(dO[e + 2], vv[2], d);
          d = fmaf(dO[e + 3], vv[3], d);
        }
      }
      ds[j] = d;
      dot = fmaf(p[j], d, dot);
    }
    float dq[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) dq[e] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      const float s = (j < N && lane < N) ? p[j] * (ds[j] - dot) : 0.f;
      ds[j] = s;
      db[j] += s;
      if (j < N) {
        const float* kr = sK + j * HD;
#pragma unroll
        for (int e = 0; e < HD; e += 4) {
          const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + e);
          dq[e] = fmaf(s, kv[0], dq[e]);
          dq[e + 1] = fmaf(s, kv[1], dq[e + 1]);
          dq[e + 2] = fmaf(s, kv[2], dq[e + 2]);
          dq[e + 3] = fmaf(s, kv[3], dq[e + 3]);
        }
      }
    }
    if (lane < N) {
#pragma unroll
      for (int j = 0; j < MAXN; ++j) {
        if (j < N) {
          sP[lane * LDP + j] = p[j];
          sDS[lane * LDP + j] = ds[j];
        }
      }
    }
    __syncthreads();
    // lane j: dK_j = scale * sum_i dS[i][j] Q_i ;  dV_j = sum_i P[i][j] dO_i
    float dk[HD], dv[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) dk[e] = dv[e] = 0.f;
    if (lane < N) {
      for (int i = 0; i < N; ++i) {
        const float s = sDS[i * LDP + lane], pp = sP[i * LDP + lane];
        const float* qr = sQ + i * HD;
        const float* dr = sDO + i * HD;
#pragma unroll
        for (int e = 0; e < HD; e += 4) {
          const f32x4 qv = *reinterpret_cast<const f32x4*>(qr + e);
          const f32x4 dvv = *reinterpret_cast<const f32x4*>(dr + e);
          dk[e] = fmaf(s, qv[0], dk[e]);
          dk[e + 1] = fmaf(s, qv[1], dk[e + 1]);
          dk[e + 2] = fmaf(s, qv[2], dk[e + 2]);
          dk[e + 3] = fmaf(s, qv[3], dk[e + 3]);
          dv[e] = fmaf(pp, dvv[0], dv[e]);
          dv[e + 1] = fmaf(pp, dvv[1], dv[e + 1]);
          dv[e + 2] = fmaf(pp, dvv[2], dv[e + 2]);
          dv[e + 3] = fmaf(pp, dvv[3], dv[e + 3]);
        }
      }
      T* drow = dqkv + (size_t)tok * 3 * g.C + head * HD;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        F8 a, bq, c;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          a.v[e] = dq[8 * k + e] * scale;
          bq.v[e] = dk[8 * k + e] * scale;
          c.v[e] = dv[8 * k + e];
        }
        store8(drow + 8 * k, a);
        store8(drow + g.C + 8 * k, bq);
        store8(drow + 2 * g.C + 8 * k, c);
      }
    }
  }
  if (lane < N && dbias_table) {
    const int ws = g.ws;
    const int qi = lane / ws, qj = lane - qi * ws;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      if (j < N) {
        const int ki = j / ws, kj = j - ki * ws;
        const int idx = (qi - ki + ws - 1) * (2 * ws - 1) + (qj - kj + ws - 1);
        atomicAdd(dbias_table + idx * g.heads + head, db[j]);
      }
    }
  }
}

// PatchMerging gather: [B][H][W][C] -> [B][H/2][W/2][4C], channel blocks ordered (h even,w even), (h odd,w even),
// (h even,w odd), (h odd,w odd).  inverse = 1 scatters a gradient back.
template <typename T>
__global__ void patch_merge_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C8,
                                   int inverse) {
  const int H2 = H / 2, W2 = W / 2;
  const int64_t total = (int64_t)B * H2 * W2 * 4 * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % C8);
    int64_t t = i / C8;
    const int blk = (int)(t % 4);
    t /= 4;
    const int w2 = (int)(t % W2);
    t /= W2;
    const int h2 = (int)(t % H2);
    const int b = (int)(t / H2);
    const int h = 2 * h2 + (blk & 1), w = 2 * w2 + (blk >> 1);
    const size_t full = ((((size_t)b * H + h) * W + w) * C8 + cc) * 8;
    if (!inverse) store8(out + i * 8, load8(in + full));
    else store8(out + full, load8(in + i * 8));
  }
}

// ============================================================================================
// MFMA path (bf16).  One wave per (window, head); the 49 tokens are padded to 64 (4 MFMA tiles).
//
// The scores are produced TRANSPOSED, S^T[j][i] = sum_e K[j][e] Q[i][e] = mfma(A = K rows, B = Q rows), so that
// a lane holds one query column i = 16*ti + (lane & 15) and its keys j = 16*tj + 4*(lane >> 4) + r in registers:
// the softmax over j is 16 in-lane values plus two cross-lane steps (xor 16, 32), and the accumulator tiles are
// directly the B operand of the next product, which contracts over j:  O^T[e][i] = sum_j V^T[e][j] P^T[j][i].
// Its A operand V^T is read from the row-major LDS tile V[j][e] with the transposing read ds_read_b64_tr_b16,
// taking for k-step s the rows j = 16*(2s + (jj >> 2)) + 4*(lane >> 4) + (jj & 3), i.e. the same permuted k
// order in which the accumulator registers enumerate j.
// The backward pass needs P and dS in both orientations (contractions over j for dQ, over i for dK / dV); both
// are recomputed with MFMAs (S^T and S, dP^T and dP: 4 x 16 MFMAs) rather than transposed through LDS.
// ============================================================================================
constexpr int AW = 4;             // waves (units) per block
constexpr int TS = 64 * HD;       // elements of one [64 tokens][32] bf16 tile
constexpr int PSTR = 68;          // row stride (elements) of the backward pass's [16 queries][64 keys] transposing buffers

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

__device__ __forceinline__ bf16x8 frag_rows(const bf16* tile, int t, int lane) {
  // A/B fragment of a row-major [64][32] tile for token tile t: row 16t + (lane & 15), k = 8*(lane >> 4) ..
  return *reinterpret_cast<const bf16x8*>(tile + (16 * t + (lane & 15)) * HD + 8 * (lane >> 4));
}

__device__ __forceinline__ bf16x8 frag_tr(const bf16* tile, int s, int te, int lane) {
  // A fragment of tile^T: row e = 16*te + (lane & 15), k-step s in the accumulator's permuted token order
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const bf16* p0 = tile + (16 * (2 * s) + 4 * g + q) * HD + 16 * te + 4 * pp;
  const bf16* p1 = p0 + 16 * HD;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p1);
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

// 16-deep operand fragment through the transposing read: from a row-major tile (row stride `stride` elements) whose
// rows are the contraction index, lane gets column 16*tc + (lane & 15) at rows 4*(lane >> 4) .. + 3
__device__ __forceinline__ s16x4 frag_tr16(const bf16* tile, int stride, int tc, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + (4 * g + q) * stride + 16 * tc + 4 * pp));
}

// B fragment for k-step s from accumulator tiles t[2s], t[2s+1] (4 values each)
__device__ __forceinline__ bf16x8 frag_from_acc(const f32x4& a, const f32x4& b) {
  bf16x8 f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f[r] = (bf16)a[r];
    f[4 + r] = (bf16)b[r];
  }
  return f;
}

struct UnitCtx {
  int b, wh, ww, head;
  bool valid;
};

// stage the [64][32] tile of `which` (0 q, 1 k, 2 v of qkv; or dout when src_is_out) for this unit: lane <-> token
__device__ __forceinline__ void stage_tile(bf16* tile, const bf16* src, size_t row_elems, int col0, int tok, bool tvalid,
                                           int lane) {
  bf16x8 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  if (tvalid) {
    const bf16* row = src + (size_t)tok * row_elems + col0;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const bf16x8*>(row + 8 * k);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) *reinterpret_cast<bf16x8*>(tile + lane * HD + 8 * k) = v[k];
}

// scale * S + bias + mask for the accumulator tile whose element r is (row rr0 + r, col cc) in (j, i) or (i, j)
// orientation; TRANSPOSED = true: rows are keys j, cols are queries i.
template <bool TRANSPOSED>
__device__ __forceinline__ float score_of(float acc, int row, int col, int N, int ws, const float* sBias,
                                          const int* sReg, const int* sRC, int shift, float scale) {
  const int i = TRANSPOSED ? col : row, j = TRANSPOSED ? row : col;
  if (j >= N) return -INFINITY;
  const int ic = i < N ? i : N - 1;
  // sRC[t] = (t / ws) * (2ws-1) + (t % ws): the bias index is sRC[i] - sRC[j] + (ws-1)*(2ws-1) + (ws-1)
  float s = acc * scale + sBias[sRC[ic] - sRC[j] + (ws - 1) * (2 * ws)];
  if (shift && sReg[ic] != sReg[j]) s -= 100.f;
  return s;
}

__device__ __attribute__((aligned(64))) const uint32_t swin_zero_src[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
typedef __attribute__((address_space(1))) const void* swin_gptr_t;
typedef __attribute__((address_space(3))) void* swin_lptr_t;
// one wave-instruction of LDS-DMA: lane l moves 16 bytes from its own global address to lds_wave_base + 16 l
__device__ __forceinline__ void glds16_swin(const void* gsrc, bf16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((swin_gptr_t)gsrc, (swin_lptr_t)lds_wave_base, 16, 0, 0);
}
// Orders a wave's own LDS traffic for the compiler.  The LDS serves one wave's instructions in issue order, so what the
// wave wrote (or what its LDS-DMA requests delivered before a vmcnt wait) is what its later reads return.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Forward.  grid = (window chunks, heads); each wave walks `wpw` windows of one head, so that what depends on the head
// only is set up once per wave: the relative-position bias (and the -inf of the padded keys) of the 64 (i, j) elements
// a lane owns sits in registers and enters as the C operand of the product that forms the scores (divided by the scale
// that multiplies the sum afterwards), and the scores are kept in units of log2(e) so that the exponential is the bare
// v_exp_f32.  The q / k / v tiles of a window arrive by LDS-DMA (12 requests per lane, no staging registers); one tile
// set per wave keeps two blocks on a CU (194 registers per lane), whose other waves cover the wait for the next
// window's tiles.  One query tile (16 queries) at a time: 16 scores, the softmax of its column, the product with V
// and the store.
__global__ __launch_bounds__(64 * AW) void window_attn_fwd_mfma_kernel(SwinGeom g, const bf16* __restrict__ qkv,
                                                                        const float* __restrict__ bias_table,
                                                                        bf16* __restrict__ out, float scale,
                                                                        int windows_total, int wpw) {
  __shared__ __attribute__((aligned(1024))) bf16 sT[AW][3][TS];  // q, k, v tiles per wave
  __shared__ float sBias[AW][176];
  __shared__ int sReg2[AW][2][64];
  __shared__ int sTok2[AW][2][64];
  __shared__ int sRC[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gq = lane >> 4, l15 = lane & 15;
  const int N = g.ws * g.ws;
  const int head = blockIdx.y;
  const int nb = (2 * g.ws - 1) * (2 * g.ws - 1);
  const int off = (g.ws - 1) * (2 * g.ws);
  if (threadIdx.x < 64) sRC[threadIdx.x] = (threadIdx.x / g.ws) * (2 * g.ws - 1) + (threadIdx.x % g.ws);
  for (int k = lane; k < nb; k += 64) sBias[wave][k] = bias_table[k * g.heads + head];
  __syncthreads();

  // a lane's element (ti, tj, r) = (query i = 16ti + l15, key j = 16tj + 4gq + r)
  f32x4 bt[4][4];
  const float scale2 = scale * 1.4426950408889634f, rscale = 1.f / scale;
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + l15, j = 16 * tj + 4 * gq + r;
        const int ic = i < N ? i : N - 1;
        bt[ti][tj][r] = j < N ? sBias[wave][sRC[ic] - sRC[j] + off] * rscale : -INFINITY;
      }

  const int w_first = (blockIdx.x * AW + wave) * wpw;
  const bool tvalid = lane < N;
  // tokens and shift-mask regions of window `it` into table set `buf`; its three operand tiles requested into the
  // wave's tile set (lane l of request k fetches 16 bytes of token 16k + l/4; rows past the window's end a zero line)
  auto request_window = [&](int it, int buf) {
    int win = w_first + it;
    if (win >= windows_total) win = windows_total - 1;
    int u = win;
    const int ww = u % g.nww;
    u /= g.nww;
    const int wh = u % g.nwh;
    const int b = u / g.nwh;
    int tok = 0, reg = 0;
    if (tvalid) {
      int ph, pw;
      tok = token_of(g, b, wh, ww, lane, ph, pw);
      reg = region_of(ph, g.H, g.ws, g.shift) * 3 + region_of(pw, g.W, g.ws, g.shift);
    }
    sReg2[wave][buf][lane] = reg;
    sTok2[wave][buf][lane] = tvalid ? tok : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = 16 * k + (lane >> 2);
      const int tk = __shfl(tok, t, 64);
      const bool ok = t < N;
      const bf16* zero = reinterpret_cast<const bf16*>(swin_zero_src);
#pragma unroll
      for (int w3 = 0; w3 < 3; ++w3)
        glds16_swin(ok ? qkv + (size_t)tk * 3 * g.C + w3 * g.C + head * HD + 8 * (lane & 3) : zero,
                    sT[wave][w3] + k * 16 * HD);
    }
  };
  const bf16* tQ = sT[wave][0];
  const bf16* tK = sT[wave][1];
  const bf16* tV = sT[wave][2];
  request_window(0, 0);
  for (int it = 0; it < wpw; ++it) {
    const int cur = it & 1;
    const bool uvalid = w_first + it < windows_total;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_lds_fence();
    const int* sRegW = sReg2[wave][cur];
    const int* sTokW = sTok2[wave][cur];
    int ri[4], rj[4][4];
    if (g.shift) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i1 = 16 * t + l15;
        ri[t] = sRegW[i1 < N ? i1 : N - 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int x = 16 * t + 4 * gq + r;
          rj[t][r] = sRegW[x < N ? x : N - 1];
        }
      }
    }
    bf16x8 kf[4], vt[2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) kf[t] = frag_rows(tK, t, lane);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int te = 0; te < 2; ++te) vt[s][te] = frag_tr(tV, s, te, lane);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const int i = 16 * ti + l15;
      const bf16x8 qf = frag_rows(tQ, ti, lane);
      f32x4 st[4];  // S^T[tj] of this query tile
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) st[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[tj], qf, bt[ti][tj], 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sc = st[tj][r] * scale2;
          if (g.shift && ri[ti] != rj[tj][r]) sc -= 100.f * 1.4426950408889634f;
          st[tj][r] = sc;
          mx = fmaxf(mx, sc);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(st[tj][r] - mx);
          st[tj][r] = e;
          sum += e;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / sum;
      f32x4 ot[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // O^T[te] of this query tile
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = frag_from_acc(st[2 * s], st[2 * s + 1]);
#pragma unroll
        for (int te = 0; te < 2; ++te) ot[te] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[s][te], pf, ot[te], 0, 0, 0);
      }
      if (uvalid && i < N) {
        bf16* orow = out + (size_t)sTokW[i] * g.C + head * HD;
#pragma unroll
        for (int te = 0; te < 2; ++te) {
          bf16x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (bf16)(ot[te][r] * inv);
          *reinterpret_cast<bf16x4*>(orow + 16 * te + 4 * gq) = v;
        }
      }
    }
    // the tile set is free once every read of it has been served (the LDS keeps a wave's instructions in order)
    wave_lds_fence();
    if (it + 1 < wpw) request_window(it + 1, cur ^ 1);
  }
}

// grid = (window chunks, heads); each wave walks `wpw` windows of one head.  Everything that does not depend on
// the window is hoisted into registers before the walk: the relative-position bias (plus the -inf of the padded
// keys) of the 64 (i, j) elements a lane owns in either orientation, and the bias-gradient accumulator of the
// 64 elements it owns in the first orientation, which is folded onto the (2ws-1)^2 bins once, after the walk.
// Per window only the shift-mask regions are looked up (20 + 20 LDS reads per lane instead of ~6 per element).
__global__ __launch_bounds__(64 * AW, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void window_attn_bwd_mfma_kernel(SwinGeom g, const bf16* __restrict__ qkv,
                                                                           const bf16* __restrict__ dout,
                                                                           const float* __restrict__ bias_table,
                                                                           bf16* __restrict__ dqkv,
                                                                           float* __restrict__ dbias_table,
                                                                           float scale, int windows_total, int wpw,
                                                                           size_t bin_slots) {
  // q, k, v, dO tiles per wave, TWICE: the tiles of the wave's next window stream in (LDS-DMA) while the current
  // window is multiplied.  128 KiB of the 160: one block per CU, which the 512 registers per wave allow anyway.
  __shared__ __attribute__((aligned(1024))) bf16 sT2[AW][2][4][TS];
  __shared__ float sBias[AW][176];
  __shared__ float sBin[176];
  // P and dS of one query tile (16 queries x 64 keys, bf16, rows padded to 136 bytes against bank conflicts), written
  // in the first orientation's register layout and read back TRANSPOSED as MFMA operands of the products that contract
  // over the queries
  __shared__ __attribute__((aligned(16))) bf16 sPS[AW][2][16 * PSTR];
  __shared__ int sReg2[AW][2][64];
  __shared__ int sTok2[AW][2][64];
  __shared__ int sRC[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gq = lane >> 4, l15 = lane & 15;
  const int N = g.ws * g.ws;
  const int head = blockIdx.y;
  const int nb = (2 * g.ws - 1) * (2 * g.ws - 1);
  const int off = (g.ws - 1) * (2 * g.ws);
  if (threadIdx.x < 64) sRC[threadIdx.x] = (threadIdx.x / g.ws) * (2 * g.ws - 1) + (threadIdx.x % g.ws);
  if (threadIdx.x < 176) sBin[threadIdx.x] = 0.f;
  for (int k = lane; k < nb; k += 64) sBias[wave][k] = bias_table[k * g.heads + head];
  __syncthreads();

  // a lane's element (ti, tj, r) = (query i = 16ti + l15, key j = 16tj + 4gq + r).  The bias enters as the C operand
  // of the product that forms the scores, divided by the scale that multiplies the sum afterwards: as an accumulator
  // input it needs no register move and no addition.  The scores are kept in units of log2(e), so that the
  // exponential is the bare v_exp_f32.
  f32x4 bt1[4][4];
  float dbacc[4][4][4];
  const float scale2 = scale * 1.4426950408889634f, rscale = 1.f / scale;
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + l15, j = 16 * tj + 4 * gq + r;
        const int ic = i < N ? i : N - 1;
        bt1[ti][tj][r] = j < N ? sBias[wave][sRC[ic] - sRC[j] + off] * rscale : -INFINITY;
        dbacc[ti][tj][r] = 0.f;
      }
  __syncthreads();

  const int w_first = (blockIdx.x * AW + wave) * wpw;
  const bool tvalid = lane < N;
  // tokens and shift-mask regions of window `it` into table set `buf`, its four operand tiles requested into tile set
  // `buf` (16 LDS-DMA instructions per lane: lane l of instruction k fetches 16 bytes of token 16k + l/4)
  auto request_window = [&](int it, int buf) {
    int win = w_first + it;
    if (win >= windows_total) win = windows_total - 1;
    int u = win;
    const int ww = u % g.nww;
    u /= g.nww;
    const int wh = u % g.nwh;
    const int b = u / g.nwh;
    int tok = 0, reg = 0;
    if (tvalid) {
      int ph, pw;
      tok = token_of(g, b, wh, ww, lane, ph, pw);
      reg = region_of(ph, g.H, g.ws, g.shift) * 3 + region_of(pw, g.W, g.ws, g.shift);
    }
    sReg2[wave][buf][lane] = reg;
    sTok2[wave][buf][lane] = tvalid ? tok : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = 16 * k + (lane >> 2);
      const int tk = __shfl(tok, t, 64);
      const bool ok = t < N;
      const bf16* zero = reinterpret_cast<const bf16*>(swin_zero_src);
#pragma unroll
      for (int w3 = 0; w3 < 3; ++w3)
        glds16_swin(ok ? qkv + (size_t)tk * 3 * g.C + w3 * g.C + head * HD + 8 * (lane & 3) : zero,
                    sT2[wave][buf][w3] + k * 16 * HD);
      glds16_swin(ok ? dout + (size_t)tk * g.C + head * HD + 8 * (lane & 3) : zero, sT2[wave][buf][3] + k * 16 * HD);
    }
  };
  request_window(0, 0);
  bool stored = false;  // the previous window wrote its gradients (a window beyond the last one does not)
  for (int it = 0; it < wpw; ++it) {
    const int cur = it & 1;
    const bool uvalid = w_first + it < windows_total;
    // In issue order the wave has outstanding: [this window's 16 tile requests] [the previous window's 24 gradient
    // stores].  vmcnt retires in order, so "at most 24 outstanding" means the tiles have landed WITHOUT waiting for
    // the stores' acknowledgements (several microseconds under load); they get the whole window to arrive, because
    // the next wait that covers them is the next window's.  Register spills only add instructions behind the
    // requests, which makes the bound conservative.
    // The count of 24 holds when every 16-row tile of the window has a live row (N > 48, i.e. 7 x 7 windows): with
    // fewer tokens the stores of an empty tile may be branched over, fewer than 24 are issued and the bound would
    // release before the tiles have landed.  Smaller windows wait for everything.
    if (stored && N > 48) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_lds_fence();
    // the next window's tiles go into the other set, whose last reader was the window before this one (the LDS serves
    // a wave's instructions in order, and every tile and table here is private to the wave: no workgroup barrier)
    if (it + 1 < wpw) request_window(it + 1, cur ^ 1);
    stored = uvalid;
    const int* sRegW = sReg2[wave][cur];
    const int* sTokW = sTok2[wave][cur];
    const bf16* tQ = sT2[wave][cur][0];
    const bf16* tK = sT2[wave][cur][1];
    const bf16* tV = sT2[wave][cur][2];
    const bf16* tD = sT2[wave][cur][3];
    // shift-mask regions of this window for the elements the lane owns (padded tokens take token N-1's)
    int ri1[4], rj1[4][4];
    if (g.shift) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i1 = 16 * t + l15;
        ri1[t] = sRegW[i1 < N ? i1 : N - 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int x = 16 * t + 4 * gq + r;
          rj1[t][r] = sRegW[x < N ? x : N - 1];
        }
      }
    }

    // ---- rows j, columns i (lane <-> query i); one column tile ti (16 queries) at a time ------------
    f32x4 dq[2][4], dv[2][4], dk[2][4];
#pragma unroll
    for (int te = 0; te < 2; ++te)
#pragma unroll
      for (int t = 0; t < 4; ++t) dq[te][t] = dv[te][t] = dk[te][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16* bufP = sPS[wave][0];
    bf16* bufS = sPS[wave][1];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const int i = 16 * ti + l15;
      const bf16x8 qf = frag_rows(tQ, ti, lane), df = frag_rows(tD, ti, lane);
      f32x4 st[4], dp[4];
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        st[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(tK, tj, lane), qf, bt1[ti][tj], 0, 0, 0);
        dp[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(tV, tj, lane), df, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sc = st[tj][r] * scale2;
          if (g.shift && ri1[ti] != rj1[tj][r]) sc -= 100.f * 1.4426950408889634f;
          st[tj][r] = sc;
          mx = fmaxf(mx, sc);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(st[tj][r] - mx);
          st[tj][r] = e;
          sum += e;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / sum;
      float dot = 0.f;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[tj][r] *= inv;  // P^T
          dot = fmaf(st[tj][r], dp[tj][r], dot);
        }
      dot += __shfl_xor(dot, 16, 64);
      dot += __shfl_xor(dot, 32, 64);
      const float live = ((i < N) && uvalid) ? 1.f : 0.f;  // padded queries / windows add nothing to the bias
      // P^T of this query tile goes to LDS as it stands (row = the lane's query, 4 consecutive keys per store) ...
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        bf16x4 pb;
#pragma unroll
        for (int r = 0; r < 4; ++r) pb[r] = (bf16)st[tj][r];
        *reinterpret_cast<bf16x4*>(bufP + l15 * PSTR + 16 * tj + 4 * gq) = pb;
      }
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = st[tj][r] * (dp[tj][r] - dot);  // dS^T (zero where P is zero: j >= N)
          st[tj][r] = ds;
          dbacc[ti][tj][r] = fmaf(live, ds, dbacc[ti][tj][r]);
        }
      // dQ^T[e][i] += sum_j K^T[e][j] dS^T[j][i]; ... and so does dS^T, in the bf16 rounding the product above uses
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 sf = frag_from_acc(st[2 * s2], st[2 * s2 + 1]);
        *reinterpret_cast<bf16x4*>(bufS + l15 * PSTR + 16 * (2 * s2) + 4 * gq) = bf16x4{sf[0], sf[1], sf[2], sf[3]};
        *reinterpret_cast<bf16x4*>(bufS + l15 * PSTR + 16 * (2 * s2 + 1) + 4 * gq) = bf16x4{sf[4], sf[5], sf[6], sf[7]};
#pragma unroll
        for (int te = 0; te < 2; ++te)
          dq[te][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(tK, s2, te, lane), sf, dq[te][ti], 0, 0, 0);
      }
      // The products that contract over the queries take P and dS of these 16 queries as B operands (k = query,
      // n = key), read back through the transposing LDS read; their A operands are dO^T and Q^T of the same 16 tokens:
      //   dV^T[e][j] += sum_i dO^T[e][i] P[i][j];   dK^T[e][j] += sum_i Q^T[e][i] dS[i][j]
      // Padded queries (i >= N) have zero rows in Q and dO and a finite P: they add nothing.
      wave_lds_fence();
      {
        s16x4 dt[2], qt[2];
#pragma unroll
        for (int te = 0; te < 2; ++te) {
          dt[te] = frag_tr16(tD + 16 * ti * HD, HD, te, lane);
          qt[te] = frag_tr16(tQ + 16 * ti * HD, HD, te, lane);
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          const s16x4 pf = frag_tr16(bufP, PSTR, tj, lane), sfj = frag_tr16(bufS, PSTR, tj, lane);
#pragma unroll
          for (int te = 0; te < 2; ++te) {
            dv[te][tj] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dt[te], pf, dv[te][tj], 0, 0, 0);
            dk[te][tj] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qt[te], sfj, dk[te][tj], 0, 0, 0);
          }
        }
      }
    }
    if (uvalid) {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        const int i = 16 * ti + l15;
        if (i >= N) continue;
        bf16* drow = dqkv + (size_t)sTokW[i] * 3 * g.C + head * HD;
#pragma unroll
        for (int te = 0; te < 2; ++te) {
          bf16x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (bf16)(dq[te][ti][r] * scale);
          *reinterpret_cast<bf16x4*>(drow + 16 * te + 4 * gq) = v;
        }
      }
    }
    if (uvalid) {
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const int j = 16 * tj + l15;
        if (j >= N) continue;
        bf16* drow = dqkv + (size_t)sTokW[j] * 3 * g.C + head * HD;
#pragma unroll
        for (int te = 0; te < 2; ++te) {
          bf16x4 vk, vv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            vk[r] = (bf16)(dk[te][tj][r] * scale);
            vv[r] = (bf16)dv[te][tj][r];
          }
          *reinterpret_cast<bf16x4*>(drow + g.C + 16 * te + 4 * gq) = vk;
          *reinterpret_cast<bf16x4*>(drow + 2 * g.C + 16 * te + 4 * gq) = vv;
        }
      }
    }
  }
  __syncthreads();
  if (dbias_table && bin_slots == 0) {
    // fold the lane-owned (i, j) sums onto the (2ws-1)^2 relative-position bins in LDS, then one global atomic
    // per bin and block
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * ti + l15, j = 16 * tj + 4 * gq + r;
          if (i < N && j < N) atomicAdd(&sBin[sRC[i] - sRC[j] + off], dbacc[ti][tj][r]);
        }
    __syncthreads();
    if ((int)threadIdx.x < nb) atomicAdd(dbias_table + threadIdx.x * g.heads + head, sBin[threadIdx.x]);
  } else if (dbias_table) {
    // Deterministic mode: the same fold in a fixed order.  Every wave parks its 64 x 64 matrix in LDS (the operand
    // tiles are free now: exactly 4 x 16 KiB), then thread t adds up bin t -- the waves in order, the (query, key)
    // pairs of that offset in order -- into the block's own copy (blockIdx.x) of the table; the copies are added in
    // block order by the caller.
    float* mat = reinterpret_cast<float*>(&sT2[0][0][0][0]) + wave * 4096;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
        *reinterpret_cast<f32x4*>(mat + (16 * ti + l15) * 64 + 16 * tj + 4 * gq) =
            f32x4{dbacc[ti][tj][0], dbacc[ti][tj][1], dbacc[ti][tj][2], dbacc[ti][tj][3]};
    __syncthreads();
    const int t = threadIdx.x;
    if (t < nb) {
      const int span = 2 * g.ws - 1;
      const int di = t / span - (g.ws - 1), dj = t % span - (g.ws - 1);  // query row / column minus key row / column
      const float* all = reinterpret_cast<const float*>(&sT2[0][0][0][0]);
      float sum = 0.f;
      for (int w = 0; w < AW; ++w)
        for (int ki = max(0, -di); ki < min(g.ws, g.ws - di); ++ki)
          for (int kj = max(0, -dj); kj < min(g.ws, g.ws - dj); ++kj)
            sum += all[w * 4096 + ((ki + di) * g.ws + kj + dj) * 64 + ki * g.ws + kj];
      dbias_table[(size_t)blockIdx.x * bin_slots + t * g.heads + head] = sum;  // the only writer of this entry
    }
  }
}

int fill_swin(SwinGeom& g, int B, int H, int W, int C, int heads, int ws, int shift, const char* who) {
  VC_CHECK_SHAPE(B > 0 && H > 0 && W > 0 && ws > 0 && ws * ws <= MAXN && H % ws == 0 && W % ws == 0,
                 "%s: window %d does not tile %dx%d (or exceeds 7x7)", who, ws, H, W);
  VC_CHECK_SHAPE(heads > 0 && C == heads * HD, "%s: head_dim must be 32 (C=%d, heads=%d)", who, C, heads);
  VC_CHECK_SHAPE(shift >= 0 && shift < ws, "%s: bad shift %d", who, shift);
  g.B = B; g.H = H; g.W = W; g.C = C; g.heads = heads; g.ws = ws; g.shift = shift;
  g.nwh = H / ws; g.nww = W / ws;
  return VC_OK;
}

}  // namespace

extern "C" {

int vc_window_attn_fwd(int dtype, const void* qkv, const float* bias_table, void* out, int B, int H, int W, int C,
                       int heads, int ws, int shift, float scale, hipStream_t stream) {
  VC_CHECK_ARG(qkv && bias_table && out, "vc_window_attn_fwd: null pointer");
  SwinGeom g;
  int rc = fill_swin(g, B, H, W, C, heads, ws, shift, "vc_window_attn_fwd");
  if (rc) return rc;
  const int units = B * g.nwh * g.nww * heads;
  if (dtype == VC_BF16) {
    // two blocks per CU (registers and LDS), each wave walking the windows of one head
    const int windows = B * g.nwh * g.nww;
    const int chunks = 2 * vc_device_cus() / heads > 0 ? 2 * vc_device_cus() / heads : 1;
    const int wpw = ceil_div(windows, AW * chunks);
    hipLaunchKernelGGL(window_attn_fwd_mfma_kernel, dim3(ceil_div(windows, AW * wpw), heads), dim3(64 * AW), 0, stream,
                       g, (const bf16*)qkv, bias_table, (bf16*)out, scale, windows, wpw);
  } else if (dtype == VC_F32)
    hipLaunchKernelGGL((window_attn_fwd_kernel<float>), dim3(units), dim3(64), 0, stream, g, (const float*)qkv,
                       bias_table, (float*)out, scale);
  else { vc_set_error("vc_window_attn_fwd: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_window_attn_fwd");
  return VC_OK;
}

// dbias_table ([(2ws-1)^2][heads] fp32) is accumulated into with atomics
int vc_window_attn_bwd(int dtype, const void* qkv, const void* dout, const float* bias_table, void* dqkv,
                       float* dbias_table, int B, int H, int W, int C, int heads, int ws, int shift, float scale,
                       hipStream_t stream) {
  VC_CHECK_ARG(qkv && dout && bias_table && dqkv, "vc_window_attn_bwd: null pointer");
  SwinGeom g;
  int rc = fill_swin(g, B, H, W, C, heads, ws, shift, "vc_window_attn_bwd");
  if (rc) return rc;
  const int windows = B * g.nwh * g.nww;
  int wpb = 1;
  while ((int64_t)ceil_div(windows, wpb) * heads > 8192) wpb *= 2;
  dim3 grid(ceil_div(windows, wpb), heads);
  if (dtype == VC_BF16) {
    // Windows per wave: one block per CU (its LDS holds one) in ONE round, so that what a block does once -- the bias
    // tables of its head into registers, the fold of the bias gradient at the end -- is paid once per CU and not once
    // per round (three rounds of 8 windows per wave at 128 frames of stage 1: 309 us; one round of 25: see below).
    const int chunks = vc_device_cus() / heads > 0 ? vc_device_cus() / heads : 1;
    const int wpw = ceil_div(windows, AW * chunks);
    dim3 mgrid(ceil_div(windows, AW * wpw), heads);
    // deterministic mode: a copy of the bias-gradient table per block column, added in order afterwards
    const size_t nbins = (size_t)(2 * ws - 1) * (2 * ws - 1) * heads;
    const bool ordered = vc_deterministic() && dbias_table;
    float* table = dbias_table;
    if (ordered) {
      table = vc_det_workspace(nbins * mgrid.x, stream, "vc_window_attn_bwd");
      if (!table) return VC_ERR_ARG;
    }
    hipLaunchKernelGGL(window_attn_bwd_mfma_kernel, mgrid, dim3(64 * AW), 0, stream, g, (const bf16*)qkv,
                       (const bf16*)dout, bias_table, (bf16*)dqkv, table, scale, windows, wpw,
                       ordered ? nbins : (size_t)0);
    if (ordered) {
      VC_LAUNCH_CHECK("vc_window_attn_bwd");
      return vc_det_reduce(dbias_table, table, nbins, (int)mgrid.x, stream);
    }
  }
  else if (dtype == VC_F32)
    hipLaunchKernelGGL((window_attn_bwd_kernel<float>), grid, dim3(64), 0, stream, g, (const float*)qkv,
                       (const float*)dout, bias_table, (float*)dqkv, dbias_table, scale, windows, wpb);
  else { vc_set_error("vc_window_attn_bwd: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_window_attn_bwd");
  return VC_OK;
}

int vc_patch_merge(int dtype, const void* in, void* out, int B, int H, int W, int C, int inverse,
                   hipStream_t stream) {
  VC_CHECK_ARG(in && out, "vc_patch_merge: null pointer");
  VC_CHECK_SHAPE(B > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "vc_patch_merge: need even H, W and C %% 8 == 0");
  const int64_t total = (int64_t)B * H * W * (C / 8);
  int64_t blocks = ceil_div64(total, 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == VC_BF16)
    hipLaunchKernelGGL((patch_merge_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)in,
                       (bf16*)out, B, H, W, C / 8, inverse);
  else if (dtype == VC_F32)
    hipLaunchKernelGGL((patch_merge_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)in,
                       (float*)out, B, H, W, C / 8, inverse);
  else { vc_set_error("vc_patch_merge: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_patch_merge");
  return VC_OK;
}

}  // extern "C"
