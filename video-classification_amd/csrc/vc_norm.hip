// BatchNorm (training mode, statistics per GROUP of rows) and LayerNorm for channels-last
// activations [M][C] on gfx950.  All kernels are HBM-bound: 16-byte (8 x bf16) or 32-byte
// (8 x fp32) accesses per lane, fp32 arithmetic.
//
// Grouping: the reference runs its frame encoders once per time step, so BatchNorm
// statistics are taken over the B frames of ONE time step and the running statistics get T
// momentum updates per forward, in time order.  Frames are laid out time-major here
// (frame = t*B + b), so a group is a contiguous block of `rpg` rows and G = T.
// Conv3D / plain batches use G = 1.
#include <stdlib.h>

#include "vc_common.h"

namespace {

constexpr int NTHR = 256;

// Cache policy of the bf16 streaming passes, two bits per kernel (1 = non-temporal loads, 2 = non-temporal stores):
// bits 0-1 forward apply, bits 2-3 backward apply, bits 4-5 backward reduce.  The apply passes touch every byte once;
// streaming them past L2 leaves the GEMM kernels on either side their operands and fewer dirty lines to write back.
// Measured on the ResNet-50 CRNN step (32 clips): 47.98 ms with the default policy everywhere, 46.32 ms with both
// apply passes non-temporal (forward alone -0.7 ms, backward alone -1.1 ms), no change from the reduce, and less gain
// when tensors below 100 / 300 MB are left cached (46.45 / 46.94 ms).  VC_BN_NT overrides the setting (0 = default
// policy everywhere), VC_BN_NT_MIN_MB leaves smaller tensors cached.
inline int bn_nt_policy(size_t tensor_bytes) {
  static const int v = [] {
    const char* e = getenv("VC_BN_NT");
    return e ? atoi(e) : 15;
  }();
  static const size_t min_bytes = [] {
    const char* e = getenv("VC_BN_NT_MIN_MB");
    return (size_t)(e ? atoi(e) : 0) << 20;
  }();
  return tensor_bytes >= min_bytes ? v : 0;
}

// Streaming forms of load8 / store8 (non-temporal: the line is not kept in L2 behind the access).  The apply passes
// touch every byte exactly once.
__device__ __forceinline__ F8 load8_nt(const bf16* p) {
  F8 r;
  const bf16x8 a = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (float)a[i];
  return r;
}
__device__ __forceinline__ F8 load8_nt(const float* p) { return load8(p); }
__device__ __forceinline__ void store8_nt(bf16* p, const F8& r) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16)r.v[i];
  __builtin_nontemporal_store(a, reinterpret_cast<bf16x8*>(p));
}
__device__ __forceinline__ void store8_nt(float* p, const F8& r) { store8(p, r); }

// per-(group, channel) sums of v1 and v2 over the rows of a group
// MODE 0: v1 = y,  v2 = y*y                       (forward statistics)
// MODE 1: v1 = dz, v2 = dz * xhat                 (backward reductions), dz = dout * (act > 0 if relu)
// WRITE_DZ (MODE 1): the ReLU decisions come from the bit mask `bits` ([row][C/8] bytes, bit i = channel 8*cc + i
// was positive; written by bn_apply_kernel) and dz is also written back over dout (dz_out aliases dout, so neither
// is __restrict__): the masked gradient is then both the shortcut's gradient and the apply pass's input.
template <typename T, int MODE, bool WRITE_DZ = false, int NT = 0>
__global__ void bn_reduce_kernel(const T* __restrict__ y, const T* dout, const T* __restrict__ act,
                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                 float* __restrict__ s1, float* __restrict__ s2, int C, int rpg, int rows_per_block,
                                 int relu, T* dz_out = nullptr, const uint8_t* __restrict__ bits = nullptr,
                                 size_t slot_stride = 0) {
  // slot_stride != 0 (deterministic mode): block x of a group adds into its own copy x of the [G][C] tables
  __shared__ float red[2][NTHR][8];
  auto ld = [](const T* p) { return (NT & 1) ? load8_nt(p) : load8(p); };
  auto st = [](T* p, const F8& v) {
    if (NT & 2) store8_nt(p, v);
    else store8(p, v);
  };
  const int C8 = C >> 3;
  const int lanesC = C8 < NTHR ? C8 : NTHR;
  const int RL = NTHR / lanesC;
  const int tid = threadIdx.x;
  const int cl = tid % lanesC, rl = tid / lanesC;
  const int g = blockIdx.y;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(rpg, r_begin + rows_per_block);
  for (int cc0 = 0; cc0 < C8; cc0 += lanesC) {  // trip count is uniform over the block (barriers inside)
    const int cc = cc0 + cl;
    const bool cvalid = cc < C8;
    float a1[8], a2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a1[i] = a2[i] = 0.f;
    if (rl < RL && cvalid) {
      float mu[8], is[8], sc[8], sh[8];
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          mu[i] = mean[g * C + cc * 8 + i];
          is[i] = invstd[g * C + cc * 8 + i];
          sc[i] = (relu && !act) ? scale[g * C + cc * 8 + i] : 0.f;
          sh[i] = (relu && !act) ? shift[g * C + cc * 8 + i] : 0.f;
        }
      }
      int r = r_begin + rl;
      if (MODE == 0) {  // four independent 16-byte loads in flight per lane
        for (; r + 3 * RL < r_end; r += 4 * RL) {
          const size_t off = ((size_t)g * rpg + r) * C + cc * 8;
          F8 yv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) yv[u] = ld(y + off + (size_t)u * RL * C);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              a1[i] += yv[u].v[i];
              a2[i] = fmaf(yv[u].v[i], yv[u].v[i], a2[i]);
            }
        }
      } else {
        for (; r + 1 * RL < r_end; r += 2 * RL) {
          const size_t off = ((size_t)g * rpg + r) * C + cc * 8;
          F8 yv[2], dv[2], av[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            yv[u] = ld(y + off + (size_t)u * RL * C);
            dv[u] = ld(dout + off + (size_t)u * RL * C);
          }
          if (WRITE_DZ) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const unsigned mb = bits[((size_t)g * rpg + r + u * RL) * C8 + cc];
#pragma unroll
              for (int i = 0; i < 8; ++i) av[u].v[i] = (mb >> i) & 1 ? 1.f : 0.f;
            }
          } else if (relu && act) {
#pragma unroll
            for (int u = 0; u < 2; ++u) av[u] = ld(act + off + (size_t)u * RL * C);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (WRITE_DZ || (relu && act)) {
#pragma unroll
              for (int i = 0; i < 8; ++i) dv[u].v[i] = av[u].v[i] > 0.f ? dv[u].v[i] : 0.f;
            } else if (relu) {
#pragma unroll
              for (int i = 0; i < 8; ++i) dv[u].v[i] = fmaf(yv[u].v[i], sc[i], sh[i]) > 0.f ? dv[u].v[i] : 0.f;
            }
            if (WRITE_DZ) st(dz_out + off + (size_t)u * RL * C, dv[u]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              a1[i] += dv[u].v[i];
              a2[i] = fmaf(dv[u].v[i], (yv[u].v[i] - mu[i]) * is[i], a2[i]);
            }
          }
        }
      }
      for (; r < r_end; r += RL) {
        const size_t off = ((size_t)g * rpg + r) * C + cc * 8;
        const F8 yv = ld(y + off);
        if (MODE == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            a1[i] += yv.v[i];
            a2[i] = fmaf(yv.v[i], yv.v[i], a2[i]);
          }
        } else {
          F8 dv = ld(dout + off);
          if (WRITE_DZ) {
            const unsigned mb = bits[((size_t)g * rpg + r) * C8 + cc];
#pragma unroll
            for (int i = 0; i < 8; ++i) dv.v[i] = (mb >> i) & 1 ? dv.v[i] : 0.f;
          } else if (relu) {
            if (act) {  // the activation had a residual added: its sign is not a function of y alone
              const F8 av = ld(act + off);
#pragma unroll
              for (int i = 0; i < 8; ++i) dv.v[i] = av.v[i] > 0.f ? dv.v[i] : 0.f;
            } else {    // recompute the ReLU decision from y (saves one activation read)
#pragma unroll
              for (int i = 0; i < 8; ++i) dv.v[i] = fmaf(yv.v[i], sc[i], sh[i]) > 0.f ? dv.v[i] : 0.f;
            }
          }
          if (WRITE_DZ) st(dz_out + off, dv);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            a1[i] += dv.v[i];
            a2[i] = fmaf(dv.v[i], (yv.v[i] - mu[i]) * is[i], a2[i]);
          }
        }
      }
    }
    // reduce over the row lanes through LDS (channel-contiguous image), then ONE coalesced atomic per
    // (statistic, channel): lane t adds channel cc0*8 + t
    const int CB = lanesC * 8;  // channels covered by this sweep
    float* r1 = &red[0][0][0];
    float* r2 = r1 + NTHR * 8;
    if (rl < RL) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        r1[rl * CB + cl * 8 + i] = a1[i];
        r2[rl * CB + cl * 8 + i] = a2[i];
      }
    }
    __syncthreads();
    for (int ch = tid; ch < CB; ch += NTHR) {
      const int cg = cc0 * 8 + ch;
      if (cg < C) {
        float t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < RL; ++k) {
          t1 += r1[k * CB + ch];
          t2 += r2[k * CB + ch];
        }
        atomicAdd(s1 + (size_t)blockIdx.x * slot_stride + g * C + cg, t1);
        atomicAdd(s2 + (size_t)blockIdx.x * slot_stride + g * C + cg, t2);
      }
    }
    __syncthreads();
  }
}

// flags: bit 0 = update the running statistics, bit 1 = zero sum/sumsq after reading them
// sums[g][c] += sum over the slabs of group g of partial[slab][0][c] (and [1] into sumsq); a block covers
// `chunk` slabs of one group and 256 channels
// mean / invstd (nullable, [G][C]): the second statistic is sum v*y and is converted to sum v*xhat on the way,
// invstd * (sum v*y - mean * sum v) (linear, so it can be done per chunk of slabs)
__global__ void bn_partial_reduce_kernel(const float* __restrict__ partial, float* __restrict__ sum,
                                         float* __restrict__ sumsq, int spg, int C, int chunk,
                                         const float* __restrict__ mean = nullptr,
                                         const float* __restrict__ invstd = nullptr, size_t slot_stride = 0) {
  // slot_stride != 0 (deterministic mode): the block of slab chunk y adds into its own copy y of the tables
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.z;
  if (c >= C) return;
  const int s0 = blockIdx.y * chunk, s1 = min(spg, s0 + chunk);
  float a = 0.f, b = 0.f;
#pragma unroll 8
  for (int sl = s0; sl < s1; ++sl) {  // unrolled: the loads of eight slabs in flight, the sums in slab order
    const float* p = partial + ((size_t)g * spg + sl) * 2 * C;
    a += p[c];
    b += p[C + c];
  }
  if (mean) b = invstd[g * C + c] * (b - mean[g * C + c] * a);
  atomicAdd(sum + (size_t)blockIdx.y * slot_stride + g * C + c, a);
  atomicAdd(sumsq + (size_t)blockIdx.y * slot_stride + g * C + c, b);
}

__global__ void bn_finalize_kernel(float* __restrict__ sum, float* __restrict__ sumsq,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   float* __restrict__ scale, float* __restrict__ shift, int G, int C, float cnt,
                                   float eps, float momentum, int flags, int64_t* __restrict__ nbt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += G;  // num_batches_tracked: one count per group (time step), as the per-step loop does
  if (c >= C) return;
  const int update_running = flags & 1;
  float rm = 0.f, rv = 0.f;
  if (update_running) {
    rm = running_mean[c];
    rv = running_var[c];
  }
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  for (int g = 0; g < G; ++g) {
    const double m = (double)sum[g * C + c] / cnt;
    double var = (double)sumsq[g * C + c] / cnt - m * m;
    if (flags & 2) {
      sum[g * C + c] = 0.f;
      sumsq[g * C + c] = 0.f;
    }
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean_out[g * C + c] = (float)m;
    invstd_out[g * C + c] = is;
    const float sc = ga * is;
    scale[g * C + c] = sc;
    shift[g * C + c] = be - (float)m * sc;
    if (update_running) {
      rm = (1.f - momentum) * rm + momentum * (float)m;
      rv = (1.f - momentum) * rv + momentum * (float)(var * (cnt / (cnt - 1.0)));
    }
  }
  if (update_running) {
    running_mean[c] = rm;
    running_var[c] = rv;
  }
}

// bn_finalize_kernel for G <= FIN_MAXG with a thread per (group, channel): the double-precision divisions and the
// square root of the G groups run side by side instead of one after the other in a channel's thread (the launch was
// 9 us for 16 groups, whatever the channel count); the running statistics are then updated by the channel's first
// thread from the fp32 values parked in LDS, in group order as before (bit-identical results).
constexpr int FIN_CH = 16, FIN_MAXG = 64;
__global__ __launch_bounds__(256) void bn_finalize_par_kernel(
    float* __restrict__ sum, float* __restrict__ sumsq, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ scale, float* __restrict__ shift,
    int G, int C, float cnt, float eps, float momentum, int flags, int64_t* __restrict__ nbt) {
  __shared__ float lm[FIN_MAXG][FIN_CH], lv[FIN_MAXG][FIN_CH];
  const int cl = threadIdx.x % FIN_CH, gl = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += G;
  const bool live = c < C;
  const float ga = (live && gamma) ? gamma[c] : 1.f, be = (live && beta) ? beta[c] : 0.f;
  for (int g = gl; g < G; g += 256 / FIN_CH) {
    if (!live) continue;
    const double m = (double)sum[g * C + c] / cnt;
    double var = (double)sumsq[g * C + c] / cnt - m * m;
    if (flags & 2) {
      sum[g * C + c] = 0.f;
      sumsq[g * C + c] = 0.f;
    }
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean_out[g * C + c] = (float)m;
    invstd_out[g * C + c] = is;
    const float sc = ga * is;
    scale[g * C + c] = sc;
    shift[g * C + c] = be - (float)m * sc;
    lm[g][cl] = (float)m;
    lv[g][cl] = (float)(var * (cnt / (cnt - 1.0)));
  }
  if (!(flags & 1)) return;  // uniform over the block
  __syncthreads();
  if (gl == 0 && live) {
    float rm = running_mean[c], rv = running_var[c];
    for (int g = 0; g < G; ++g) {
      rm = (1.f - momentum) * rm + momentum * lm[g][cl];
      rv = (1.f - momentum) * rv + momentum * lv[g][cl];
    }
    running_mean[c] = rm;
    running_var[c] = rv;
  }
}

// scale/shift from running statistics (eval mode), G = 1
__global__ void bn_eval_params_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float* __restrict__ scale, float* __restrict__ shift, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = (gamma ? gamma[c] : 1.f) * rsqrtf(running_var[c] + eps);
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - running_mean[c] * sc;
}

// Row-streaming layout shared by the apply kernels: a block owns `rows_per_block` rows of ONE group and every
// channel; a thread owns an 8-channel chunk (its per-channel coefficients stay in registers) and walks rows
// rl, rl+RL, ... four at a time so that several 16-byte loads are in flight per lane.
struct RowWalk {
  int lanesC, RL, cl, rl, r_begin, r_end;
};
__device__ __forceinline__ RowWalk row_walk(int C8, int rpg, int rows_per_block) {
  RowWalk w;
  w.lanesC = C8 < NTHR ? C8 : NTHR;
  w.RL = NTHR / w.lanesC;
  w.cl = threadIdx.x % w.lanesC;
  w.rl = threadIdx.x / w.lanesC;
  w.r_begin = blockIdx.x * rows_per_block;
  w.r_end = min(rpg, w.r_begin + rows_per_block);
  if (w.rl >= w.RL) w.r_end = w.r_begin;  // threads beyond lanesC*RL idle (C8 not a divisor of the block)
  return w;
}

// bit i = element i is positive (the ReLU decisions of 8 channels in one byte)
__device__ __forceinline__ uint8_t positive_bits(const F8& o) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) m |= (o.v[i] > 0.f ? 1u : 0u) << i;
  return (uint8_t)m;
}

// out = act(y * scale[g] + shift[g] (+ residual)); bits (nullable): [row][C/8] ReLU decisions of out
template <typename T, int NT = 0>
__global__ __launch_bounds__(NTHR) void bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const T* __restrict__ residual, T* __restrict__ out, int C,
                                                        int rpg, int rows_per_block, int relu,
                                                        uint8_t* __restrict__ bits = nullptr) {
  const int C8 = C >> 3, g = blockIdx.y;
  const RowWalk w = row_walk(C8, rpg, rows_per_block);
  for (int cc = w.cl; cc < C8; cc += w.lanesC) {
    const F8 sc = load8(scale + (size_t)g * C + cc * 8), sh = load8(shift + (size_t)g * C + cc * 8);
    const size_t base = (size_t)g * rpg * C + cc * 8;
    const size_t rstride = (size_t)w.RL * C;
    int r = w.r_begin + w.rl;
    for (; r + 3 * w.RL < w.r_end; r += 4 * w.RL) {
      const size_t off = base + (size_t)r * C;
      F8 yv[4], rv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) yv[u] = (NT & 1) ? load8_nt(y + off + u * rstride) : load8(y + off + u * rstride);
      if (residual) {
#pragma unroll
        for (int u = 0; u < 4; ++u) rv[u] = (NT & 1) ? load8_nt(residual + off + u * rstride) : load8(residual + off + u * rstride);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        F8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = fmaf(yv[u].v[i], sc.v[i], sh.v[i]);
        if (residual) {
#pragma unroll
          for (int i = 0; i < 8; ++i) o.v[i] += rv[u].v[i];
        }
        if (relu) {
#pragma unroll
          for (int i = 0; i < 8; ++i) o.v[i] = fmaxf(o.v[i], 0.f);
        }
        if (NT & 2) store8_nt(out + off + u * rstride, o);
        else store8(out + off + u * rstride, o);
        if (bits) bits[(off + u * rstride) >> 3] = positive_bits(o);
      }
    }
    for (; r < w.r_end; r += w.RL) {
      const size_t off = base + (size_t)r * C;
      const F8 yv = (NT & 1) ? load8_nt(y + off) : load8(y + off);
      F8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o.v[i] = fmaf(yv.v[i], sc.v[i], sh.v[i]);
      if (residual) {
        const F8 rv = (NT & 1) ? load8_nt(residual + off) : load8(residual + off);
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] += rv.v[i];
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = fmaxf(o.v[i], 0.f);
      }
      if (NT & 2) store8_nt(out + off, o);
      else store8(out + off, o);
      if (bits) bits[off >> 3] = positive_bits(o);
    }
  }
}

// dy = scale * (dz - s1/cnt - xhat * s2/cnt) = ka*dz + kb*y + kc per (group, channel); optionally dres = dz
template <typename T, int NT = 0>
__global__ __launch_bounds__(NTHR) void bn_bwd_apply_kernel(
    const T* __restrict__ y, const T* __restrict__ dout, const T* __restrict__ act, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ s1, const float* __restrict__ s2, T* __restrict__ dy, T* __restrict__ dres, int C,
    int rpg, int rows_per_block, float inv_cnt, int relu) {
  const int C8 = C >> 3, g = blockIdx.y;
  const RowWalk w = row_walk(C8, rpg, rows_per_block);
  const bool recompute = relu && !act;
  for (int cc = w.cl; cc < C8; cc += w.lanesC) {
    const size_t po = (size_t)g * C + cc * 8;
    const F8 sc = load8(scale + po);
    F8 sh, kb, kc;
    {
      const F8 mu = load8(mean + po), is = load8(invstd + po), a1 = load8(s1 + po), a2 = load8(s2 + po);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float t = sc.v[i] * is.v[i] * a2.v[i] * inv_cnt;
        kb.v[i] = -t;
        kc.v[i] = fmaf(t, mu.v[i], -sc.v[i] * a1.v[i] * inv_cnt);
        sh.v[i] = 0.f;
      }
      if (recompute) sh = load8(shift + po);
    }
    const size_t base = (size_t)g * rpg * C + cc * 8;
    const size_t rstride = (size_t)w.RL * C;
    auto one = [&](size_t off, const F8& yv, F8 dv, const F8& av) {
      if (relu) {
        if (recompute) {
#pragma unroll
          for (int i = 0; i < 8; ++i) dv.v[i] = fmaf(yv.v[i], sc.v[i], sh.v[i]) > 0.f ? dv.v[i] : 0.f;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) dv.v[i] = av.v[i] > 0.f ? dv.v[i] : 0.f;
        }
      }
      if (dres) {
        if (NT & 2) store8_nt(dres + off, dv);
        else store8(dres + off, dv);
      }
      F8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o.v[i] = fmaf(sc.v[i], dv.v[i], fmaf(kb.v[i], yv.v[i], kc.v[i]));
      if (NT & 2) store8_nt(dy + off, o);
      else store8(dy + off, o);
    };
    int r = w.r_begin + w.rl;
    for (; r + 3 * w.RL < w.r_end; r += 4 * w.RL) {
      const size_t off = base + (size_t)r * C;
      F8 yv[4], dv[4], av[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        yv[u] = (NT & 1) ? load8_nt(y + off + u * rstride) : load8(y + off + u * rstride);
        dv[u] = (NT & 1) ? load8_nt(dout + off + u * rstride) : load8(dout + off + u * rstride);
      }
      if (relu && !recompute) {
#pragma unroll
        for (int u = 0; u < 4; ++u) av[u] = (NT & 1) ? load8_nt(act + off + u * rstride) : load8(act + off + u * rstride);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(off + u * rstride, yv[u], dv[u], av[u]);
    }
    for (; r < w.r_end; r += w.RL) {
      const size_t off = base + (size_t)r * C;
      const F8 yv = load8(y + off), dv = load8(dout + off);
      F8 av = yv;
      if (relu && !recompute) av = load8(act + off);
      one(off, yv, dv, av);
    }
  }
}

// ---------------------------------------------------------------------------
// BatchNorm + ReLU + max-pool in one pass each way (ResNet stem: 3x3 / 2 / 1; Conv3D: (1,2,2) after every block).
// The full-resolution activation relu(bn(y)) is never written: forward reads y and writes the pooled tensor and
// the arg-max taps; backward gathers the pooled gradient through those taps where the unfused path would have
// read a full-resolution gradient tensor (written once, read twice).
// Rows are frames n of H x W pixels; a BatchNorm group is `npg` consecutive frames.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void bn_pool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                   const float* __restrict__ shift, T* __restrict__ out, uint8_t* __restrict__ idx,
                                   int N, int H, int W, int C8, int Ho, int Wo, int k, int s, int p, int npg) {
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % C8);
    int64_t t = i / C8;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const size_t po = (size_t)(n / npg) * C8 * 8 + cc * 8;
    const F8 sc = load8(scale + po), sh = load8(shift + po);
    float best[8];
    int bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      best[j] = -INFINITY;
      bi[j] = 0;
    }
    for (int r = 0; r < k; ++r) {
      const int ih = oh * s - p + r;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int q = 0; q < k; ++q) {
        const int iw = ow * s - p + q;
        if ((unsigned)iw >= (unsigned)W) continue;
        const F8 v = load8(y + ((((size_t)n * H + ih) * W + iw) * C8 + cc) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // rounded to the storage type before the comparison: the same winner as a max-pool over the stored
          // activation would pick
          const float a = to_f32(from_f32<T>(fmaxf(fmaf(v.v[j], sc.v[j], sh.v[j]), 0.f)));
          if (a > best[j]) {
            best[j] = a;
            bi[j] = r * k + q;
          }
        }
      }
    }
    F8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = best[j];
    store8(out + i * 8, o);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo |= (uint32_t)bi[j] << (8 * j);
      hi |= (uint32_t)bi[4 + j] << (8 * j);
    }
    *reinterpret_cast<uint2*>(idx + i * 8) = make_uint2(lo, hi);
  }
}

// gradient that reaches element (n, ih, iw, channels 8cc..8cc+7) of the max-pool's input: the pooled gradients of
// the windows that contain it and whose recorded tap is this element
template <typename T>
__device__ __forceinline__ F8 pool_grad_gather(const T* __restrict__ dyp, const uint8_t* __restrict__ idx, int n,
                                               int ih, int iw, int cc, int C8, int Ho, int Wo, int k, int s, int p) {
  F8 acc;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc.v[j] = 0.f;
  const int oh_lo = max(0, (ih + p - k + s) / s), oh_hi = min(Ho - 1, (ih + p) / s);
  const int ow_lo = max(0, (iw + p - k + s) / s), ow_hi = min(Wo - 1, (iw + p) / s);
  for (int oh = oh_lo; oh <= oh_hi; ++oh) {
    const int r = ih - (oh * s - p);
    if (r < 0 || r >= k) continue;
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
      const int q = iw - (ow * s - p);
      if (q < 0 || q >= k) continue;
      const size_t o = ((((size_t)n * Ho + oh) * Wo + ow) * C8 + cc) * 8;
      const uint2 w = *reinterpret_cast<const uint2*>(idx + o);
      const F8 d = load8(dyp + o);
      const int tap = r * k + q;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int b = (int)(((j < 4 ? w.x : w.y) >> (8 * (j & 3))) & 0xff);
        if (b == tap) acc.v[j] += d.v[j];
      }
    }
  }
  return acc;
}

// MODE 0: s1[g][c] += sum dz, s2[g][c] += sum dz * xhat     (dz = gathered gradient where bn(y) > 0)
// MODE 1: dy = scale * (dz - s1/cnt - xhat * s2/cnt)
template <typename T, int MODE>
__global__ __launch_bounds__(NTHR) void bn_pool_bwd_kernel(
    const T* __restrict__ y, const T* __restrict__ dyp, const uint8_t* __restrict__ idx,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ s1, float* __restrict__ s2, T* __restrict__ dy, int C, int H,
    int W, int Ho, int Wo, int k, int s, int p, int rpg, int rows_per_block, float inv_cnt, size_t slot_stride) {
  __shared__ float red[2][NTHR][8];
  const int C8 = C >> 3, g = blockIdx.y;
  const RowWalk w = row_walk(C8, rpg, rows_per_block);
  const int HW = H * W;
  for (int cc0 = 0; cc0 < C8; cc0 += w.lanesC) {  // uniform trip count (barriers inside for MODE 0)
    const int cc = cc0 + w.cl;
    const bool cvalid = cc < C8;
    float a1[8], a2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a1[i] = a2[i] = 0.f;
    if (cvalid) {
      const size_t po = (size_t)g * C + cc * 8;
      const F8 sc = load8(scale + po), sh = load8(shift + po), mu = load8(mean + po), is = load8(invstd + po);
      F8 kb, kc;
      if (MODE == 1) {
        const F8 t1 = load8(s1 + po), t2 = load8(s2 + po);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float t = sc.v[i] * is.v[i] * t2.v[i] * inv_cnt;
          kb.v[i] = -t;
          kc.v[i] = fmaf(t, mu.v[i], -sc.v[i] * t1.v[i] * inv_cnt);
        }
      }
      for (int r = w.r_begin + w.rl; r < w.r_end; r += w.RL) {
        const int pix = g * rpg + r;
        const int n = pix / HW, rem = pix - n * HW;
        const int ih = rem / W, iw = rem - ih * W;
        const size_t off = (size_t)pix * C + cc * 8;
        const F8 yv = load8(y + off);
        F8 dz = pool_grad_gather(dyp, idx, n, ih, iw, cc, C8, Ho, Wo, k, s, p);
#pragma unroll
        for (int i = 0; i < 8; ++i) dz.v[i] = fmaf(yv.v[i], sc.v[i], sh.v[i]) > 0.f ? dz.v[i] : 0.f;
        if (MODE == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            a1[i] += dz.v[i];
            a2[i] = fmaf(dz.v[i], (yv.v[i] - mu.v[i]) * is.v[i], a2[i]);
          }
        } else {
          F8 o;
#pragma unroll
          for (int i = 0; i < 8; ++i) o.v[i] = fmaf(sc.v[i], dz.v[i], fmaf(kb.v[i], yv.v[i], kc.v[i]));
          store8(dy + off, o);
        }
      }
    }
    if (MODE == 0) {
      const int CB = w.lanesC * 8;
      float* r1 = &red[0][0][0];
      float* r2 = r1 + NTHR * 8;
      if (w.rl < w.RL) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          r1[w.rl * CB + w.cl * 8 + i] = a1[i];
          r2[w.rl * CB + w.cl * 8 + i] = a2[i];
        }
      }
      __syncthreads();
      for (int ch = threadIdx.x; ch < CB; ch += NTHR) {
        const int cg = cc0 * 8 + ch;
        if (cg < C) {
          float t1 = 0.f, t2 = 0.f;
          for (int q = 0; q < w.RL; ++q) {
            t1 += r1[q * CB + ch];
            t2 += r2[q * CB + ch];
          }
          atomicAdd(s1 + (size_t)blockIdx.x * slot_stride + g * C + cg, t1);
          atomicAdd(s2 + (size_t)blockIdx.x * slot_stride + g * C + cg, t2);
        }
      }
      __syncthreads();
    }
  }
}

// dgamma[c] (+)= sum_g s2[g][c]; dbeta[c] (+)= sum_g s1[g][c]
// flags: bit 0 = add to dgamma/dbeta instead of overwriting, bit 1 = zero s1/s2 after reading them
__global__ void bn_bwd_params_kernel(float* __restrict__ s1, float* __restrict__ s2, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int G, int C, int flags) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int g = 0; g < G; ++g) {
    a += s2[g * C + c];
    b += s1[g * C + c];
    if (flags & 2) {
      s1[g * C + c] = 0.f;
      s2[g * C + c] = 0.f;
    }
  }
  if (dgamma) dgamma[c] = (flags & 1) ? dgamma[c] + a : a;
  if (dbeta) dbeta[c] = (flags & 1) ? dbeta[c] + b : b;
}

// bn_bwd_params_kernel for G <= FIN_MAXG with a thread per (group, channel): the loads of the G groups are issued side
// by side; the channel's first thread adds them up from LDS in group order (the same sums as the sequential kernel).
__global__ __launch_bounds__(256) void bn_bwd_params_par_kernel(float* __restrict__ s1, float* __restrict__ s2,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int G, int C, int flags) {
  __shared__ float l1[FIN_MAXG][FIN_CH], l2[FIN_MAXG][FIN_CH];
  const int cl = threadIdx.x % FIN_CH, gl = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  const bool live = c < C;
  for (int g = gl; g < G; g += 256 / FIN_CH) {
    if (!live) continue;
    l1[g][cl] = s1[g * C + c];
    l2[g][cl] = s2[g * C + c];
    if (flags & 2) {
      s1[g * C + c] = 0.f;
      s2[g * C + c] = 0.f;
    }
  }
  __syncthreads();
  if (gl == 0 && live) {
    float a = 0.f, b = 0.f;
    for (int g = 0; g < G; ++g) {
      a += l2[g][cl];
      b += l1[g][cl];
    }
    if (dgamma) dgamma[c] = (flags & 1) ? dgamma[c] + a : a;
    if (dbeta) dbeta[c] = (flags & 1) ? dbeta[c] + b : b;
  }
}

// ------------------------------ LayerNorm ---------------------------------
// A row is handled by LPR lanes (16, 32 or 64 = one wave), so that narrow rows (Swin stage 1: C = 96 = 12
// 16-byte chunks) still keep most lanes busy: a wave normalises 64/LPR rows at once.  C % 8 == 0.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int LPR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, T* __restrict__ out,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int M, int C,
                                                      float eps) {
  constexpr int RPB = 256 / LPR;  // rows per block
  const int sub = threadIdx.x % LPR;
  const int row = blockIdx.x * RPB + threadIdx.x / LPR;
  const bool live = row < M;      // dead rows still take part in the shuffles
  const int C8 = C >> 3;
  const T* xr = x + (size_t)(live ? row : 0) * C;
  float s = 0.f, q = 0.f;
  for (int cc = sub; cc < C8; cc += LPR) {
    const F8 v = load8(xr + cc * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s += v.v[i];
      q = fmaf(v.v[i], v.v[i], q);
    }
  }
  const float mu = group_sum<LPR>(s) / C;
  // two-pass variance on the values just read (they are in cache): exact for constant rows
  float d2 = 0.f;
  for (int cc = sub; cc < C8; cc += LPR) {
    const F8 v = load8(xr + cc * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) d2 = fmaf(v.v[i] - mu, v.v[i] - mu, d2);
  }
  const float rs = rsqrtf(group_sum<LPR>(d2) / C + eps);
  (void)q;
  if (!live) return;
  if (sub == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
  for (int cc = sub; cc < C8; cc += LPR) {
    const F8 v = load8(xr + cc * 8), ga = load8(gamma + cc * 8), be = load8(beta + cc * 8);
    F8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = fmaf((v.v[i] - mu) * rs, ga.v[i], be.v[i]);
    store8(out + (size_t)row * C + cc * 8, o);
  }
}

// dx per row; dgamma/dbeta: per-thread partial sums over the rows the thread sees (its channel chunks are
// fixed), folded in LDS per block and added to global memory with one atomic per channel and block.
// NCH = channel chunks per lane (C8 <= LPR * NCH).
template <typename T, int LPR, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dout,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, T* __restrict__ dx,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                      int C, int rows_per_block, const T* __restrict__ add,
                                                      size_t slot_stride) {
  // slot_stride != 0 (deterministic mode): the row groups of a block fold their sums one after the other instead of
  // with LDS atomics, and the block adds into its own copy of dgamma / dbeta (copy blockIdx.x)
  extern __shared__ float lds[];  // [2][C]
  constexpr int RPB = 256 / LPR;
  const int C8 = C >> 3;
  const int sub = threadIdx.x % LPR, rsl = threadIdx.x / LPR;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float pg[NCH][8], pb[NCH][8];
  F8 ga[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int cc = sub + LPR * k;
#pragma unroll
    for (int i = 0; i < 8; ++i) pg[k][i] = pb[k][i] = 0.f;
    if (cc < C8) ga[k] = load8(gamma + cc * 8);
  }
  // The rows of the NEXT row group are requested before the current one is worked on (PF: up to two chunks per lane;
  // beyond that the registers do not hold two groups): a lane has two batches of loads in flight instead of one, which
  // matters where a launch is 1-2 blocks per CU (the 25088 x 384 tensors of Swin stage 3).
  constexpr bool PF = NCH <= 2;
  F8 xn[NCH], dn[NCH], an[NCH];
  float mun = 0.f, rsn = 0.f;
  auto fetch = [&](int rb) {
    const int row = rb + rsl;
    const bool lv = row < r1;
    const size_t ro = (size_t)(lv ? row : r0) * C;
    mun = mean[lv ? row : r0];
    rsn = rstd[lv ? row : r0];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int cc = sub + LPR * k;
      if (cc < C8) {
        xn[k] = load8(x + ro + cc * 8);
        dn[k] = load8(dout + ro + cc * 8);
        if (add) an[k] = load8(add + ro + cc * 8);
      }
    }
  };
  if (PF && r0 < r1) fetch(r0);
  for (int rb = r0; rb < r1; rb += RPB) {  // uniform trip count: every lane joins the shuffles
    const int row = rb + rsl;
    const bool live = row < r1;
    const size_t ro = (size_t)(live ? row : r0) * C;
    if (!PF) fetch(rb);
    const float mu = mun, rs = rsn;
    F8 xv[NCH], dv[NCH], av[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      xv[k] = xn[k];
      dv[k] = dn[k];
      av[k] = an[k];
    }
    if (PF && rb + RPB < r1) fetch(rb + RPB);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int cc = sub + LPR * k;
      if (cc < C8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xv[k].v[i] - mu) * rs, dg = dv[k].v[i] * ga[k].v[i];
          xv[k].v[i] = xh;
          a += dg;
          b = fmaf(dg, xh, b);
          if (live) {
            pg[k][i] = fmaf(dv[k].v[i], xh, pg[k][i]);
            pb[k][i] += dv[k].v[i];
          }
        }
      }
    }
    a = group_sum<LPR>(a) / C;
    b = group_sum<LPR>(b) / C;
    if (live) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int cc = sub + LPR * k;
        if (cc < C8) {
          F8 o;
#pragma unroll
          for (int i = 0; i < 8; ++i) o.v[i] = rs * (dv[k].v[i] * ga[k].v[i] - a - xv[k].v[i] * b);
          if (add) {  // gradient that reaches x past the normalisation (the residual stream)
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] += av[k].v[i];
          }
          store8(dx + ro + cc * 8, o);
        }
      }
    }
  }
  if (slot_stride) {
    // every row group parks its sums in its own [2][C] slice (the launch reserves RPB of them), then thread i adds
    // entry i of the slices in row-group order
    float* mine = lds + (size_t)rsl * 2 * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int cc = sub + LPR * k;
      if (cc < C8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          mine[cc * 8 + i] = pg[k][i];
          mine[C + cc * 8 + i] = pb[k][i];
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
      float t = lds[i];
      for (int r = 1; r < RPB; ++r) t += lds[(size_t)r * 2 * C + i];
      lds[i] = t;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int cc = sub + LPR * k;
      if (cc < C8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          atomicAdd(&lds[cc * 8 + i], pg[k][i]);
          atomicAdd(&lds[C + cc * 8 + i], pb[k][i]);
        }
      }
    }
    __syncthreads();
  }
  dgamma += (size_t)blockIdx.x * slot_stride;
  dbeta += (size_t)blockIdx.x * slot_stride;
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    atomicAdd(dgamma + i, lds[i]);
    atomicAdd(dbeta + i, lds[C + i]);
  }
}

inline int bn_reduce_blocks() {
  static const int v = [] {
    const char* e = getenv("VC_BN_REDUCE_BLOCKS");
    const int n = e ? atoi(e) : 0;
    return n > 0 ? n : 512;
  }();
  return v;
}

template <typename T, int MODE, bool WRITE_DZ = false>
int launch_reduce(const void* y, const void* dout, const void* act, const float* mean, const float* invstd,
                  const float* scale, const float* shift, float* s1, float* s2, int G, int rpg, int C, int relu,
                  hipStream_t stream, void* dz_out = nullptr, const uint8_t* bits = nullptr) {
  // about four blocks per CU: enough loads in flight to stream at HBM rate, few enough that the atomics
  // that combine the blocks stay cheap; at least 16 rows per thread
  const int C8 = C / 8, lanesC = C8 < NTHR ? C8 : NTHR, RL = NTHR / lanesC;
  int rows_per_block = 16 * RL;
  while (rows_per_block < rpg && (int64_t)ceil_div(rpg, rows_per_block) * G > bn_reduce_blocks()) rows_per_block *= 2;
  dim3 grid(ceil_div(rpg, rows_per_block), G);
  if (vc_deterministic() && grid.x > 1) {
    // every block adds into its own copy of the two tables; the copies are then added in block order
    const size_t n = (size_t)G * C;
    float* ws = vc_det_workspace(2 * n * grid.x, stream, "BatchNorm reduction");
    if (!ws) return VC_ERR_ARG;
    hipLaunchKernelGGL((bn_reduce_kernel<T, MODE, WRITE_DZ, 0>), grid, dim3(NTHR), 0, stream, (const T*)y,
                       (const T*)dout, (const T*)act, mean, invstd, scale, shift, ws, ws + n * grid.x, C, rpg,
                       rows_per_block, relu, (T*)dz_out, bits, n);
    vc_det_reduce(s1, ws, n, (int)grid.x, stream);
    return vc_det_reduce(s2, ws + n * grid.x, n, (int)grid.x, stream);
  }
#define VC_BN_REDUCE(NT_)                                                                                     \
  hipLaunchKernelGGL((bn_reduce_kernel<T, MODE, WRITE_DZ, NT_>), grid, dim3(NTHR), 0, stream, (const T*)y,    \
                     (const T*)dout, (const T*)act, mean, invstd, scale, shift, s1, s2, C, rpg, rows_per_block, \
                     relu, (T*)dz_out, bits, (size_t)0)
  switch (MODE == 1 && sizeof(T) == 2 ? (bn_nt_policy((size_t)G * rpg * C * 2) >> 4) & 3 : 0) {
    case 1: VC_BN_REDUCE(1); break;
    case 2: VC_BN_REDUCE(2); break;
    case 3: VC_BN_REDUCE(3); break;
    default: VC_BN_REDUCE(0); break;
  }
#undef VC_BN_REDUCE
  return VC_OK;
}

// rows per block of the apply kernels: a multiple of one unrolled sweep (4*RL rows), sized for >= ~4096 blocks
inline int apply_rows_per_block(int G, int rpg, int C) {
  const int C8 = C / 8, lanesC = C8 < NTHR ? C8 : NTHR, RL = NTHR / lanesC;
  const int sweep = 4 * RL;
  int rows = sweep;
  while (rows < 8 * sweep && (int64_t)ceil_div(rpg, rows * 2) * G >= 4096) rows *= 2;
  return rows;
}

// sum / sumsq [G][C] += the slab sums of `partial`, a block per (256 channels, chunk of slabs, group); deterministic
// mode: every chunk adds into its own copy of the tables, the copies are added in chunk order afterwards
int launch_partial_reduce(const float* partial, float* sum, float* sumsq, int G, int spg, int C, const float* mean,
                          const float* invstd, hipStream_t stream) {
  int chunk = 16;
  while (ceil_div(spg, chunk) * G * ceil_div(C, 256) > 2048) chunk *= 2;
  const dim3 grid(ceil_div(C, 256), ceil_div(spg, chunk), G);
  if (vc_deterministic() && grid.y > 1) {
    const size_t n = (size_t)G * C;
    float* ws = vc_det_workspace(2 * n * grid.y, stream, "BatchNorm slab sums");
    if (!ws) return VC_ERR_ARG;
    hipLaunchKernelGGL(bn_partial_reduce_kernel, grid, dim3(256), 0, stream, partial, ws, ws + n * grid.y, spg, C, chunk,
                       mean, invstd, n);
    vc_det_reduce(sum, ws, n, (int)grid.y, stream);
    return vc_det_reduce(sumsq, ws + n * grid.y, n, (int)grid.y, stream);
  }
  hipLaunchKernelGGL(bn_partial_reduce_kernel, grid, dim3(256), 0, stream, partial, sum, sumsq, spg, C, chunk, mean,
                     invstd, (size_t)0);
  return VC_OK;
}

}  // namespace

extern "C" {

// s1/s2 [G][C] must be zeroed by the caller (they are accumulated with atomics)
int vc_bn_stats(int dtype, const void* y, float* sum, float* sumsq, int G, int rpg, int C, hipStream_t stream) {
  VC_CHECK_ARG(y && sum && sumsq, "vc_bn_stats: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0 && G > 0 && rpg > 0, "vc_bn_stats: need C %% 8 == 0 (C=%d)", C);
  int rc = VC_OK;
  if (dtype == VC_BF16)
    rc = launch_reduce<bf16, 0>(y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sum, sumsq, G, rpg, C, 0, stream);
  else if (dtype == VC_F32)
    rc = launch_reduce<float, 0>(y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sum, sumsq, G, rpg, C, 0, stream);
  else { vc_set_error("vc_bn_stats: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  if (rc) return rc;
  VC_LAUNCH_CHECK("vc_bn_stats");
  return VC_OK;
}

// partial: [G * spg slabs][2][C] written by vc_conv_fwd_stats (a slab is 64 consecutive rows, rpg = 64 * spg);
// sum/sumsq [G][C] are accumulated into (the caller keeps them zeroed, as for vc_bn_stats)
int vc_bn_stats_from_partial(const float* partial, float* sum, float* sumsq, int G, int spg, int C,
                             hipStream_t stream) {
  VC_CHECK_ARG(partial && sum && sumsq, "vc_bn_stats_from_partial: null pointer");
  VC_CHECK_SHAPE(G > 0 && G < 65536 && spg > 0 && C > 0, "vc_bn_stats_from_partial: bad shape");
  if (int rc = launch_partial_reduce(partial, sum, sumsq, G, spg, C, nullptr, nullptr, stream)) return rc;
  VC_LAUNCH_CHECK("vc_bn_stats_from_partial");
  return VC_OK;
}

// BatchNorm backward reductions from the slab sums of vc_conv_dgrad_bn_tail (partial[slab][0][c] = sum dz,
// [slab][1][c] = sum dz*y over 64 rows): s1[g][c] += sum dz, s2[g][c] += sum dz * xhat.  s1/s2 zeroed by the caller.
int vc_bn_bwd_sums_from_partial(const float* partial, const float* mean, const float* invstd, float* s1, float* s2,
                                int G, int spg, int C, hipStream_t stream) {
  VC_CHECK_ARG(partial && mean && invstd && s1 && s2, "vc_bn_bwd_sums_from_partial: null pointer");
  VC_CHECK_SHAPE(G > 0 && G < 65536 && spg > 0 && C > 0, "vc_bn_bwd_sums_from_partial: bad shape");
  if (int rc = launch_partial_reduce(partial, s1, s2, G, spg, C, mean, invstd, stream)) return rc;
  VC_LAUNCH_CHECK("vc_bn_bwd_sums_from_partial");
  return VC_OK;
}

int vc_bn_finalize_count(float* sum, float* sumsq, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* mean, float* invstd, float* scale, float* shift, int G, int C,
                         int rpg, float eps, float momentum, int flags, int64_t* num_batches_tracked,
                         hipStream_t stream);

int vc_bn_finalize(float* sum, float* sumsq, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float* mean, float* invstd, float* scale, float* shift, int G, int C, int rpg,
                   float eps, float momentum, int flags, hipStream_t stream) {
  return vc_bn_finalize_count(sum, sumsq, gamma, beta, running_mean, running_var, mean, invstd, scale, shift, G, C, rpg,
                              eps, momentum, flags, nullptr, stream);
}

// vc_bn_finalize that also adds G to the module's num_batches_tracked (device int64, nullable) in the same launch
int vc_bn_finalize_count(float* sum, float* sumsq, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* mean, float* invstd, float* scale, float* shift, int G, int C,
                         int rpg, float eps, float momentum, int flags, int64_t* num_batches_tracked,
                         hipStream_t stream) {
  VC_CHECK_ARG(sum && sumsq && mean && invstd && scale && shift, "vc_bn_finalize: null pointer");
  VC_CHECK_ARG(!(flags & 1) || (running_mean && running_var), "vc_bn_finalize: running stats missing");
  if (G > 1 && G <= FIN_MAXG)
    hipLaunchKernelGGL(bn_finalize_par_kernel, dim3(ceil_div(C, FIN_CH)), dim3(256), 0, stream, sum, sumsq, gamma, beta,
                       running_mean, running_var, mean, invstd, scale, shift, G, C, (float)rpg, eps, momentum, flags,
                       num_batches_tracked);
  else
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, stream, sum, sumsq, gamma, beta,
                       running_mean, running_var, mean, invstd, scale, shift, G, C, (float)rpg, eps, momentum, flags,
                       num_batches_tracked);
  VC_LAUNCH_CHECK("vc_bn_finalize");
  return VC_OK;
}

int vc_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float* scale, float* shift, int C, float eps, hipStream_t stream) {
  VC_CHECK_ARG(running_mean && running_var && scale && shift, "vc_bn_eval_params: null pointer");
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, stream, gamma, beta, running_mean,
                     running_var, scale, shift, C, eps);
  VC_LAUNCH_CHECK("vc_bn_eval_params");
  return VC_OK;
}

int vc_bn_apply_mask(int dtype, const void* y, const float* scale, const float* shift, const void* residual, void* out,
                     uint8_t* bits, int G, int rpg, int C, int relu, hipStream_t stream);

int vc_bn_apply(int dtype, const void* y, const float* scale, const float* shift, const void* residual, void* out,
                int G, int rpg, int C, int relu, hipStream_t stream) {
  return vc_bn_apply_mask(dtype, y, scale, shift, residual, out, nullptr, G, rpg, C, relu, stream);
}

// vc_bn_apply that also records the ReLU decisions: bits[row][C/8], bit i of a byte = out[row][8*cc + i] > 0
int vc_bn_apply_mask(int dtype, const void* y, const float* scale, const float* shift, const void* residual, void* out,
                     uint8_t* bits, int G, int rpg, int C, int relu, hipStream_t stream) {
  VC_CHECK_ARG(y && scale && shift && out, "vc_bn_apply: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0, "vc_bn_apply: need C %% 8 == 0 (C=%d)", C);
  VC_CHECK_SHAPE(G > 0 && G < 65536 && rpg > 0, "vc_bn_apply: bad group count %d", G);
  const int rows = apply_rows_per_block(G, rpg, C);
  const dim3 grid(ceil_div(rpg, rows), G);
  if (dtype == VC_BF16) {
    switch (bn_nt_policy((size_t)G * rpg * C * 2) & 3) {
#define VC_BN_APPLY(NT_)                                                                                     \
  hipLaunchKernelGGL((bn_apply_kernel<bf16, NT_>), grid, dim3(NTHR), 0, stream, (const bf16*)y, scale, shift, \
                     (const bf16*)residual, (bf16*)out, C, rpg, rows, relu, bits)
      case 1: VC_BN_APPLY(1); break;
      case 2: VC_BN_APPLY(2); break;
      case 3: VC_BN_APPLY(3); break;
      default: VC_BN_APPLY(0); break;
#undef VC_BN_APPLY
    }
  } else if (dtype == VC_F32)
    hipLaunchKernelGGL((bn_apply_kernel<float>), grid, dim3(NTHR), 0, stream, (const float*)y, scale, shift,
                       (const float*)residual, (float*)out, C, rpg, rows, relu, bits);
  else { vc_set_error("vc_bn_apply: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_bn_apply");
  return VC_OK;
}

// s1/s2 [G][C] must be zeroed by the caller.  With relu set, `act` (the activation output) gives the ReLU mask;
// act == NULL recomputes it as y*scale+shift > 0 (valid when no residual was added before the ReLU).
int vc_bn_bwd_reduce(int dtype, const void* y, const void* dout, const void* act, const float* mean,
                     const float* invstd, const float* scale, const float* shift, float* s1, float* s2, int G, int rpg,
                     int C, int relu, hipStream_t stream) {
  VC_CHECK_ARG(y && dout && mean && invstd && s1 && s2 && (!relu || act || (scale && shift)),
               "vc_bn_bwd_reduce: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0, "vc_bn_bwd_reduce: need C %% 8 == 0 (C=%d)", C);
  int rc = VC_OK;
  if (dtype == VC_BF16)
    rc = launch_reduce<bf16, 1>(y, dout, act, mean, invstd, scale, shift, s1, s2, G, rpg, C, relu, stream);
  else if (dtype == VC_F32)
    rc = launch_reduce<float, 1>(y, dout, act, mean, invstd, scale, shift, s1, s2, G, rpg, C, relu, stream);
  else { vc_set_error("vc_bn_bwd_reduce: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  if (rc) return rc;
  VC_LAUNCH_CHECK("vc_bn_bwd_reduce");
  return VC_OK;
}

// vc_bn_bwd_reduce for an activation out = relu(bn(y) + residual): the ReLU mask comes from the bit mask that
// vc_bn_apply_mask recorded and the masked gradient dz = dout * mask is written back over dout.  dz is the
// residual's gradient as it stands, and vc_bn_bwd_apply then runs on it with relu = 0 (the activation is never
// read again, no second gradient tensor is written).
int vc_bn_bwd_reduce_mask(int dtype, const void* y, void* dout, const uint8_t* bits, const float* mean,
                          const float* invstd, float* s1, float* s2, int G, int rpg, int C, hipStream_t stream) {
  VC_CHECK_ARG(y && dout && bits && mean && invstd && s1 && s2, "vc_bn_bwd_reduce_mask: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0, "vc_bn_bwd_reduce_mask: need C %% 8 == 0 (C=%d)", C);
  int rc = VC_OK;
  if (dtype == VC_BF16)
    rc = launch_reduce<bf16, 1, true>(y, dout, nullptr, mean, invstd, nullptr, nullptr, s1, s2, G, rpg, C, 1, stream, dout,
                                      bits);
  else if (dtype == VC_F32)
    rc = launch_reduce<float, 1, true>(y, dout, nullptr, mean, invstd, nullptr, nullptr, s1, s2, G, rpg, C, 1, stream,
                                       dout, bits);
  else { vc_set_error("vc_bn_bwd_reduce_mask: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  if (rc) return rc;
  VC_LAUNCH_CHECK("vc_bn_bwd_reduce_mask");
  return VC_OK;
}

int vc_bn_bwd_apply(int dtype, const void* y, const void* dout, const void* act, const float* mean,
                    const float* invstd, const float* scale, const float* shift, const float* s1, const float* s2,
                    void* dy, void* dres, int G, int rpg, int C, int relu, hipStream_t stream) {
  VC_CHECK_ARG(y && dout && mean && invstd && scale && s1 && s2 && dy && (!relu || act || shift),
               "vc_bn_bwd_apply: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0, "vc_bn_bwd_apply: need C %% 8 == 0 (C=%d)", C);
  VC_CHECK_SHAPE(G > 0 && G < 65536 && rpg > 0, "vc_bn_bwd_apply: bad group count %d", G);
  const float inv_cnt = 1.f / (float)rpg;
  const int rows = apply_rows_per_block(G, rpg, C);
  const dim3 grid(ceil_div(rpg, rows), G);
  if (dtype == VC_BF16) {
    switch ((bn_nt_policy((size_t)G * rpg * C * 2) >> 2) & 3) {
#define VC_BN_BWD_APPLY(NT_)                                                                                          \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, NT_>), grid, dim3(NTHR), 0, stream, (const bf16*)y, (const bf16*)dout, \
                     (const bf16*)act, mean, invstd, scale, shift, s1, s2, (bf16*)dy, (bf16*)dres, C, rpg, rows,     \
                     inv_cnt, relu)
      case 1: VC_BN_BWD_APPLY(1); break;
      case 2: VC_BN_BWD_APPLY(2); break;
      case 3: VC_BN_BWD_APPLY(3); break;
      default: VC_BN_BWD_APPLY(0); break;
#undef VC_BN_BWD_APPLY
    }
  } else if (dtype == VC_F32)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), grid, dim3(NTHR), 0, stream, (const float*)y, (const float*)dout,
                       (const float*)act, mean, invstd, scale, shift, s1, s2, (float*)dy, (float*)dres, C, rpg, rows,
                       inv_cnt, relu);
  else { vc_set_error("vc_bn_bwd_apply: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_bn_bwd_apply");
  return VC_OK;
}

// relu(y * scale[g] + shift[g]) max-pooled k x k / s / p per frame, in one pass: y [N][H][W][C] (N frames, groups of
// npg consecutive frames) -> out [N][Ho][Wo][C] and the arg-max taps idx (one byte per output element).
int vc_bn_pool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, int N,
                   int H, int W, int C, int k, int s, int p, int npg, hipStream_t stream) {
  VC_CHECK_ARG(y && scale && shift && out && idx, "vc_bn_pool_fwd: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0 && N > 0 && H > 0 && W > 0 && k > 0 && k <= 15 && s > 0 && p >= 0 && p < k && npg > 0,
                 "vc_bn_pool_fwd: bad shape");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  VC_CHECK_SHAPE(Ho > 0 && Wo > 0, "vc_bn_pool_fwd: empty output");
  const int64_t total = (int64_t)N * Ho * Wo * (C / 8);
  const unsigned grid = (unsigned)(ceil_div64(total, NTHR) < 65536 * 16 ? ceil_div64(total, NTHR) : 65536 * 16);
  if (dtype == VC_BF16)
    hipLaunchKernelGGL((bn_pool_fwd_kernel<bf16>), dim3(grid), dim3(NTHR), 0, stream, (const bf16*)y, scale, shift,
                       (bf16*)out, idx, N, H, W, C / 8, Ho, Wo, k, s, p, npg);
  else if (dtype == VC_F32)
    hipLaunchKernelGGL((bn_pool_fwd_kernel<float>), dim3(grid), dim3(NTHR), 0, stream, (const float*)y, scale, shift,
                       (float*)out, idx, N, H, W, C / 8, Ho, Wo, k, s, p, npg);
  else { vc_set_error("vc_bn_pool_fwd: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
  VC_LAUNCH_CHECK("vc_bn_pool_fwd");
  return VC_OK;
}

// Backward of vc_bn_pool_fwd.  phase 0: s1 / s2 [G][C] (zeroed by the caller) += the BatchNorm backward reductions of
// dz = (pooled gradient dyp gathered through idx) * (bn(y) > 0); phase 1: dy = the gradient of the BatchNorm's input.
int vc_bn_pool_bwd(int dtype, int phase, const void* y, const void* dyp, const uint8_t* idx, const float* mean,
                   const float* invstd, const float* scale, const float* shift, float* s1, float* s2, void* dy, int N,
                   int H, int W, int C, int k, int s, int p, int G, hipStream_t stream) {
  VC_CHECK_ARG(y && dyp && idx && mean && invstd && scale && shift && s1 && s2 && (phase == 0 || dy),
               "vc_bn_pool_bwd: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0 && N > 0 && G > 0 && G < 65536 && N % G == 0 && k > 0 && s > 0 && p >= 0 && p < k,
                 "vc_bn_pool_bwd: bad shape");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  const int rpg = N / G * H * W;
  const float inv_cnt = 1.f / (float)rpg;
  const int C8 = C / 8, lanesC = C8 < NTHR ? C8 : NTHR, RL = NTHR / lanesC;
  int rows = 16 * RL;
  if (phase == 0) {
    while (rows < rpg && (int64_t)ceil_div(rpg, rows) * G > bn_reduce_blocks()) rows *= 2;
  } else {
    rows = 4 * RL;
    while (rows < 32 * RL && (int64_t)ceil_div(rpg, rows * 2) * G >= 4096) rows *= 2;
  }
  const dim3 grid(ceil_div(rpg, rows), G);
  // deterministic mode, phase 0: a copy of the two tables per block, added in block order afterwards
  const bool ordered = phase == 0 && vc_deterministic() && grid.x > 1;
  const size_t n = (size_t)G * C;
  float* const s1_user = s1;
  float* const s2_user = s2;
  size_t slot_stride = 0;
  if (ordered) {
    float* ws = vc_det_workspace(2 * n * grid.x, stream, "vc_bn_pool_bwd");
    if (!ws) return VC_ERR_ARG;
    s1 = ws;
    s2 = ws + n * grid.x;
    slot_stride = n;
  }
#define VC_BN_POOL_BWD(T, MODE)                                                                                        \
  hipLaunchKernelGGL((bn_pool_bwd_kernel<T, MODE>), grid, dim3(NTHR), 0, stream, (const T*)y, (const T*)dyp, idx, mean, \
                     invstd, scale, shift, s1, s2, (T*)dy, C, H, W, Ho, Wo, k, s, p, rpg, rows, inv_cnt, slot_stride)
  if (dtype == VC_BF16) {
    if (phase == 0) VC_BN_POOL_BWD(bf16, 0); else VC_BN_POOL_BWD(bf16, 1);
  } else if (dtype == VC_F32) {
    if (phase == 0) VC_BN_POOL_BWD(float, 0); else VC_BN_POOL_BWD(float, 1);
  } else { vc_set_error("vc_bn_pool_bwd: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
#undef VC_BN_POOL_BWD
  VC_LAUNCH_CHECK("vc_bn_pool_bwd");
  if (ordered) {
    vc_det_reduce(s1_user, s1, n, (int)grid.x, stream);
    return vc_det_reduce(s2_user, s2, n, (int)grid.x, stream);
  }
  return VC_OK;
}

int vc_bn_bwd_params(float* s1, float* s2, float* dgamma, float* dbeta, int G, int C, int flags,
                     hipStream_t stream) {
  VC_CHECK_ARG(s1 && s2, "vc_bn_bwd_params: null pointer");
  if (G > 1 && G <= FIN_MAXG)
    hipLaunchKernelGGL(bn_bwd_params_par_kernel, dim3(ceil_div(C, FIN_CH)), dim3(256), 0, stream, s1, s2, dgamma, dbeta,
                       G, C, flags);
  else
    hipLaunchKernelGGL(bn_bwd_params_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, stream, s1, s2, dgamma, dbeta, G, C,
                       flags);
  VC_LAUNCH_CHECK("vc_bn_bwd_params");
  return VC_OK;
}

int vc_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* out, float* mean,
                     float* rstd, int M, int C, float eps, hipStream_t stream) {
  VC_CHECK_ARG(x && gamma && beta && out && mean && rstd, "vc_layernorm_fwd: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0 && M > 0, "vc_layernorm_fwd: need C %% 8 == 0 (C=%d)", C);
  const int lpr = C <= 128 ? 16 : (C <= 256 ? 32 : 64);
  dim3 grid(ceil_div(M, 256 / lpr));
#define VC_LN_FWD(T, LPR)                                                                                        \
  hipLaunchKernelGGL((ln_fwd_kernel<T, LPR>), grid, dim3(256), 0, stream, (const T*)x, gamma, beta, (T*)out, mean, \
                     rstd, M, C, eps)
  if (dtype == VC_BF16) {
    if (lpr == 16) VC_LN_FWD(bf16, 16); else if (lpr == 32) VC_LN_FWD(bf16, 32); else VC_LN_FWD(bf16, 64);
  } else if (dtype == VC_F32) {
    if (lpr == 16) VC_LN_FWD(float, 16); else if (lpr == 32) VC_LN_FWD(float, 32); else VC_LN_FWD(float, 64);
  } else { vc_set_error("vc_layernorm_fwd: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
#undef VC_LN_FWD
  VC_LAUNCH_CHECK("vc_layernorm_fwd");
  return VC_OK;
}

int vc_layernorm_bwd_add(int dtype, const void* x, const void* dout, const float* gamma, const float* mean,
                         const float* rstd, const void* add, void* dx, float* dgamma, float* dbeta, int M, int C,
                         hipStream_t stream);

// dgamma/dbeta are accumulated into (caller zeroes or keeps running gradient)
int vc_layernorm_bwd(int dtype, const void* x, const void* dout, const float* gamma, const float* mean,
                     const float* rstd, void* dx, float* dgamma, float* dbeta, int M, int C, hipStream_t stream) {
  return vc_layernorm_bwd_add(dtype, x, dout, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, M, C, stream);
}

// vc_layernorm_bwd with dx = (gradient through the normalisation) + add (nullable, same shape and type as x; may
// alias dx): the gradient of a tensor that feeds both a LayerNorm and the residual stream, in one pass
int vc_layernorm_bwd_add(int dtype, const void* x, const void* dout, const float* gamma, const float* mean,
                         const float* rstd, const void* add, void* dx, float* dgamma, float* dbeta, int M, int C,
                         hipStream_t stream) {
  VC_CHECK_ARG(x && dout && gamma && mean && rstd && dx && dgamma && dbeta, "vc_layernorm_bwd: null pointer");
  VC_CHECK_SHAPE(C % 8 == 0 && C <= 2048 && M > 0, "vc_layernorm_bwd: need C %% 8 == 0 and C <= 2048 (C=%d)", C);
  // lanes per row / chunks per lane: (16,1) C<=128, (32,1) C<=256, (64,1) C<=512, (64,2) C<=1024, (64,4) C<=2048
  const int lpr = C <= 128 ? 16 : (C <= 256 ? 32 : 64);
  // Blocks: at most `target`, of at least 64 rows.  A block ends with a fold of its dgamma / dbeta sums (LDS atomics of
  // every lane, then 2 C global atomics), which costs more than the latency that more blocks would hide: with blocks
  // of 16 rows (1568 - 6272 blocks per launch) the Swin-RNN step went 15.97 -> 16.56 ms.
  static const int target = [] {
    const char* e = getenv("VC_LN_BWD_BLOCKS");
    const int n = e ? atoi(e) : 0;
    return n > 0 ? n : 1024;
  }();
  int rows_per_block = 64;
  while (ceil_div(M, rows_per_block) > target) rows_per_block *= 2;
  dim3 grid(ceil_div(M, rows_per_block));
  // deterministic mode: a copy of dgamma / dbeta per block, added in block order afterwards (and one [2][C] slice of
  // LDS per row group of a block: 256 / lpr of them, at most 24 KiB)
  const bool ordered = vc_deterministic();
  const size_t lds = 2 * (size_t)C * sizeof(float) * (ordered ? 256 / lpr : 1);
  float* const dg_user = dgamma;
  float* const db_user = dbeta;
  size_t slot_stride = 0;
  if (ordered) {
    float* ws = vc_det_workspace(2 * (size_t)C * grid.x, stream, "vc_layernorm_bwd");
    if (!ws) return VC_ERR_ARG;
    dgamma = ws;
    dbeta = ws + (size_t)C * grid.x;
    slot_stride = (size_t)C;
  }
#define VC_LN_BWD(T, LPR, NCH)                                                                                  \
  hipLaunchKernelGGL((ln_bwd_kernel<T, LPR, NCH>), grid, dim3(256), lds, stream, (const T*)x, (const T*)dout,   \
                     gamma, mean, rstd, (T*)dx, dgamma, dbeta, M, C, rows_per_block, (const T*)add, slot_stride)
#define VC_LN_BWD_ALL(T)                                       \
  if (lpr == 16) VC_LN_BWD(T, 16, 1);                          \
  else if (lpr == 32) VC_LN_BWD(T, 32, 1);                     \
  else if (C <= 512) VC_LN_BWD(T, 64, 1);                      \
  else if (C <= 1024) VC_LN_BWD(T, 64, 2);                     \
  else VC_LN_BWD(T, 64, 4)
  if (dtype == VC_BF16) { VC_LN_BWD_ALL(bf16); }
  else if (dtype == VC_F32) { VC_LN_BWD_ALL(float); }
  else { vc_set_error("vc_layernorm_bwd: unsupported dtype %d", dtype); return VC_ERR_DTYPE; }
#undef VC_LN_BWD_ALL
#undef VC_LN_BWD
  VC_LAUNCH_CHECK("vc_layernorm_bwd");
  if (ordered) {
    vc_det_reduce(dg_user, dgamma, (size_t)C, (int)grid.x, stream);
    return vc_det_reduce(db_user, dbeta, (size_t)C, (int)grid.x, stream);
  }
  return VC_OK;
}

}  // extern "C"
