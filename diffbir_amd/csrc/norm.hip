// GroupNorm(+SiLU), LayerNorm and row softmax for NHWC 16-bit activations (HBM-bound kernels, f32 statistics).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// GroupNorm, two launches, deterministic (no float atomics: bit-reproducible and batch-size independent):
//  1. gn_stats: grid (nchunk, B). Thread owns one 8-channel vector (16 B loads, coalesced along C) and walks
//               the chunk's rows; per-channel (sum, sumsq) are combined through LDS in a fixed order, then reduced
//               channels -> groups -> partial[b][chunk][2*groups].
//  2. gn_apply: grid (nblk, B). Every block first sums the nchunk partials of its sample in a fixed order
//               (<= 64 x 64 floats, L2 resident) -> mean / rstd per group in LDS, folds them with gamma / beta into
//               per-thread coefficients, then streams y = act(x*a + s).
// Numerics: the sums are SHIFTED — sum (x - p_g), sum (x - p_g)^2 with the pivot p_g = x[b, row 0, first channel of
// group g] (any constant gives the exact variance in exact arithmetic; one near the data removes the cancellation of
// E[x^2] - mean^2 when |mean| >> std, e.g. large-offset VAE activations over groups of ~4M elements; PyTorch's
// group_norm, which the reference calls, uses a Welford-type update for the same reason).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void gn_stats(const u16* __restrict__ x, long long ldx, float* __restrict__ partial, int HW, int C,
                         int groups, int rows_per_chunk, int rpi) {
  extern __shared__ float sh[];  // [rpi][2*C] per row-subset partial sums, then [2*C] per-channel sums
  const int CV = C >> 3;
  const int t = threadIdx.x;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cv = t % CV, rsub = t / CV;
  const int r_begin = chunk * rows_per_chunk;
  const int r_end = min(HW, r_begin + rows_per_chunk);
  float s[8], ss[8], pv[8];
  {
    const int cpg_ = C / groups;
    const u16* x0 = x + (long long)b * HW * ldx;   // row 0 of this sample
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] = ss[e] = 0.f;
      pv[e] = T::to_f32(x0[((cv * 8 + e) / cpg_) * cpg_]);
    }
  }
  const u16* xb = x + (long long)b * HW * ldx + cv * 8;
  int r = r_begin + rsub;
  for (; r + 3 * rpi < r_end; r += 4 * rpi) {  // 4 independent 16-byte loads in flight per thread
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(xb + (long long)(r + u * rpi) * ldx);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[8];
      unpack8<T>(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dl = f[e] - pv[e];
        s[e] += dl;
        ss[e] += dl * dl;
      }
    }
  }
  for (; r < r_end; r += rpi) {
    const uint4 v = *reinterpret_cast<const uint4*>(xb + (long long)r * ldx);
    float f[8];
    unpack8<T>(v, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dl = f[e] - pv[e];
      s[e] += dl;
      ss[e] += dl * dl;
    }
  }
  float* mine = sh + (long long)rsub * 2 * C;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mine[cv * 8 + e] = s[e];
    mine[C + cv * 8 + e] = ss[e];
  }
  __syncthreads();
  for (int i = t; i < 2 * C; i += blockDim.x) {
    float a = 0.f;
    for (int k = 0; k < rpi; ++k) a += sh[(long long)k * 2 * C + i];
    sh[i] = a;  // row-subset 0's slot: only this thread reads column i of the other subsets
  }
  __syncthreads();
  const int cpg = C / groups;
  float* out = partial + ((long long)b * gridDim.x + chunk) * 2 * groups;
  for (int g = t; g < 2 * groups; g += blockDim.x) {
    const int base = g < groups ? g * cpg : C + (g - groups) * cpg;
    float a = 0.f;
    for (int c = 0; c < cpg; ++c) a += sh[base + c];
    out[g] = a;
  }
}

template <typename T>
__global__ void gn_apply(const u16* __restrict__ x, long long ldx, u16* __restrict__ y, long long ldy,
                         const float* __restrict__ partial, const float* __restrict__ gamma,
                         const float* __restrict__ beta, int nchunk, int HW, int C, int groups, float eps,
                         int rows_per_blk, int rpi, int silu, const float* __restrict__ ext) {
  __shared__ float stat[128];  // mean[groups], rstd[groups]
  const int CV = C >> 3;
  const int t = threadIdx.x;
  const int b = blockIdx.y;
  const int cpg = C / groups;
  if (ext) {   // statistics supplied by the caller (tiled VAE: aggregated over tiles): [B][mean(groups) | var(groups)]
    if (t < groups) {
      stat[t] = ext[(long long)b * 2 * groups + t];
      stat[groups + t] = rsqrtf(ext[(long long)b * 2 * groups + groups + t] + eps);
    }
    __syncthreads();
  } else {
  if (t < 2 * groups) {
    const float* pb = partial + (long long)b * nchunk * 2 * groups + t;
    float a = 0.f;
#pragma unroll 8
    for (int k = 0; k < nchunk; ++k) a += pb[(long long)k * 2 * groups];
    stat[t] = a;
  }
  __syncthreads();
  float mean = 0.f, rstd = 0.f;
  if (t < groups) {
    const float n = (float)HW * (float)cpg;
    const float ms = stat[t] / n;                                    // mean of the shifted data
    const float var = fmaxf(stat[groups + t] / n - ms * ms, 0.f);
    mean = ms + T::to_f32(x[(long long)b * HW * ldx + t * cpg]);     // + the pivot gn_stats subtracted
    rstd = rsqrtf(var + eps);
  }
  __syncthreads();
  if (t < groups) {
    stat[t] = mean;
    stat[groups + t] = rstd;
  }
  __syncthreads();
  }  // own statistics
  const int cv = t % CV, rsub = t / CV;
  float av[8], sv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cv * 8 + e;
    const int g = c / cpg;
    const float a = stat[groups + g] * gamma[c];
    av[e] = a;
    sv[e] = beta[c] - stat[g] * a;
  }
  const int r_begin = blockIdx.x * rows_per_blk;
  const int r_end = min(HW, r_begin + rows_per_blk);
  const u16* xb = x + (long long)b * HW * ldx + cv * 8;
  u16* yb = y + (long long)b * HW * ldy + cv * 8;
  for (int r = r_begin + rsub; r < r_end; r += rpi) {
    const uint4 v = *reinterpret_cast<const uint4*>(xb + (long long)r * ldx);
    float f[8];
    unpack8<T>(v, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float o = f[e] * av[e] + sv[e];
      f[e] = silu ? silu_f(o) : o;
    }
    *reinterpret_cast<uint4*>(yb + (long long)r * ldy) = pack8<T>(f);
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave64 per group of RW rows (RW independent load -> reduce -> normalise chains in flight per
// wave: the kernel is latency-bound otherwise), rows held in registers, two-pass statistics in f32.
// ------------------------------------------------------------------------------------------------
template <typename T, int MAXV, int RW>
__global__ __launch_bounds__(256) void ln_kernel(const u16* __restrict__ x, long long ldx, u16* __restrict__ y,
                                                 long long ldy, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, int rows, int C, int Cpad,
                                                 float eps) {
  const int lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= rows) return;
  const int NV = Cpad >> 3;
  float f[RW][MAXV][8];
  float sum[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    sum[r] = 0.f;
    const long long row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int v = lane + 64 * k;
      if (v < NV) {
        const uint4 q = *reinterpret_cast<const uint4*>(x + row * ldx + v * 8);
        unpack8<T>(q, f[r][k]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (v * 8 + e >= C) f[r][k][e] = 0.f;
          sum[r] += f[r][k][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[r][k][e] = 0.f;
      }
    }
  }
  float mean[RW], rstd[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) mean[r] = wave_sum(sum[r]) / (float)C;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int v = lane + 64 * k;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dlt = (v * 8 + e < C) ? f[r][k][e] - mean[r] : 0.f;
        sq += dlt * dlt;
      }
    }
    sum[r] = sq;
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) rstd[r] = rsqrtf(wave_sum(sum[r]) / (float)C + eps);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int v = lane + 64 * k;
    if (v < NV) {
      float gm[8], bt[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = v * 8 + e;
        gm[e] = c < C ? gamma[c] : 0.f;
        bt[e] = c < C ? beta[c] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        if (row0 + r < rows) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (v * 8 + e < C) ? (f[r][k][e] - mean[r]) * rstd[r] * gm[e] + bt[e] : 0.f;
          *reinterpret_cast<uint4*>(y + (row0 + r) * ldy + v * 8) = pack8<T>(o);
        }
      }
    }
  }
}

// LayerNorm, lane-group variant: LPR lanes share a row (64 / LPR rows per wave), each lane holds CPL 16-byte chunks
// (chunk index j + LPR*k), so all 64 lanes load even for narrow rows (C = 320 is 40 chunks: the wave-per-row kernel
// above leaves 24 of 64 lanes idle there) and every load instruction covers whole 128-byte segments of 64/LPR rows.
// Two-pass f32 statistics with xor-shuffle reductions inside the lane group.
template <typename T, int LPR, int CPL>
__global__ __launch_bounds__(256) void ln2_kernel(const u16* __restrict__ x, long long ldx, u16* __restrict__ y,
                                                  long long ldy, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int rows, int C, float eps) {
  constexpr int RPW = 64 / LPR;  // rows per wave
  const int lane = threadIdx.x & 63;
  const int j = lane % LPR;
  const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool rok = row < rows;
  const long long rr = rok ? row : rows - 1;
  float f[CPL][8];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int v = j + LPR * k;
    const uint4 q = *reinterpret_cast<const uint4*>(x + rr * ldx + v * 8);
    unpack8<T>(q, f[k]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (v * 8 + e >= C) f[k][e] = 0.f;
      sum += f[k][e];
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int v = j + LPR * k;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dlt = (v * 8 + e < C) ? f[k][e] - mean : 0.f;
      sq += dlt * dlt;
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = rsqrtf(sq / (float)C + eps);
  if (!rok) return;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int v = j + LPR * k;
    float o8[8], gm[8], bt[8];
    if (v * 8 + 8 <= C) {  // whole chunk inside C: 16-byte loads of the affine parameters (L1-resident)
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + v * 8), g1 = *reinterpret_cast<const float4*>(gamma + v * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(beta + v * 8), b1 = *reinterpret_cast<const float4*>(beta + v * 8 + 4);
      gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
      bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w; bt[4] = b1.x; bt[5] = b1.y; bt[6] = b1.z; bt[7] = b1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = v * 8 + e;
        gm[e] = c < C ? gamma[c] : 0.f;
        bt[e] = c < C ? beta[c] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (v * 8 + e < C) ? (f[k][e] - mean) * rstd * gm[e] + bt[e] : 0.f;
    *reinterpret_cast<uint4*>(y + row * ldy + v * 8) = pack8<T>(o8);
  }
}

// ------------------------------------------------------------------------------------------------
// Row softmax in place (one 256-thread block per row).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_kernel(u16* __restrict__ x, long long ld, int L) {
  __shared__ float sh[8];
  u16* row = x + (long long)blockIdx.x * ld;
  const int t = threadIdx.x;
  const int NV = (int)(ld >> 3);
  float mx = -1e30f;
  for (int v = t; v < NV; v += 256) {
    float f[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(row + v * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (v * 8 + e < L) mx = fmaxf(mx, f[e]);
  }
  mx = block_reduce(mx, true, sh);
  float sum = 0.f;
  for (int v = t; v < NV; v += 256) {
    float f[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(row + v * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (v * 8 + e < L) sum += __expf(f[e] - mx);
  }
  sum = block_reduce(sum, false, sh);
  const float inv = 1.f / sum;
  for (int v = t; v < NV; v += 256) {
    float f[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(row + v * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (v * 8 + e < L) ? __expf(f[e] - mx) * inv : 0.f;
    *reinterpret_cast<uint4*>(row + v * 8) = pack8<T>(f);
  }
}

}  // namespace

namespace {
// partial sums of gn_stats -> mean / biased variance per (sample, group): [B][mean(groups) | var(groups)]
template <typename T>
__global__ void gn_finalize(const u16* __restrict__ x, long long ldx, const float* __restrict__ partial,
                            float* __restrict__ out, int nchunk, int HW, int C, int groups) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= groups) return;
  const int cpg = C / groups;
  const float* pb = partial + (long long)b * nchunk * 2 * groups;
  float a = 0.f, q = 0.f;
  for (int k = 0; k < nchunk; ++k) {
    a += pb[(long long)k * 2 * groups + t];
    q += pb[(long long)k * 2 * groups + groups + t];
  }
  const float n = (float)HW * (float)cpg;
  const float ms = a / n;
  out[(long long)b * 2 * groups + t] = ms + T::to_f32(x[(long long)b * HW * ldx + t * cpg]);
  out[(long long)b * 2 * groups + groups + t] = fmaxf(q / n - ms * ms, 0.f);
}
}  // namespace

namespace {
// partial sums of gn_stats -> per (sample, channel) affine of the normalisation: ab[b][0][c] = rstd_g * gamma[c],
// ab[b][1][c] = beta[c] - mean_g * rstd_g * gamma[c]  (y = x * a + s; consumed by the fused transformer head, xformer.hip)
template <typename T>
__global__ void gn_finalize_affine(const u16* __restrict__ x, long long ldx, const float* __restrict__ partial,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ ab, int nchunk, int HW, int C, int groups, float eps) {
  __shared__ float stat[128];
  const int b = blockIdx.x, t = threadIdx.x;
  const int cpg = C / groups;
  if (t < groups) {
    const float* pb = partial + (long long)b * nchunk * 2 * groups;
    float a = 0.f, q = 0.f;
    for (int k = 0; k < nchunk; ++k) {
      a += pb[(long long)k * 2 * groups + t];
      q += pb[(long long)k * 2 * groups + groups + t];
    }
    const float n = (float)HW * (float)cpg;
    const float ms = a / n;
    stat[t] = ms + T::to_f32(x[(long long)b * HW * ldx + t * cpg]);
    stat[groups + t] = rsqrtf(fmaxf(q / n - ms * ms, 0.f) + eps);
  }
  __syncthreads();
  for (int c = t; c < C; c += blockDim.x) {
    const int g = c / cpg;
    const float a = stat[groups + g] * gamma[c];
    ab[(long long)b * 2 * C + c] = a;
    ab[(long long)b * 2 * C + C + c] = beta[c] - stat[g] * a;
  }
}
}  // namespace

namespace {
// GroupNorm statistics from the column sums emitted by dbir_gemm epilogues (dbir_gemm_desc.stats): p[tile][2][N] with `rows`
// rows per tile; the normalised tensor is [producer 1 columns | producer 2 columns].
__global__ __launch_bounds__(256) void gn_from_partials(const float* __restrict__ p1, int N1, const float* __restrict__ p2,
                                                        int N2, int rows, int HW, int groups, float eps,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ mean_var, float* __restrict__ scale_shift) {
  // one wave per (sample, group), four groups per block: lane l sums the (tile, channel) cells l, l + 64, ... of its group in
  // f64 (the per-tile sums are f32), then a butterfly over the wave — the first version walked a group with a few threads
  // of ONE block per sample and took 7 us, as long as the statistics pass it replaces
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (g >= groups) return;
  const int C = N1 + N2, cpg = C / groups, tiles = HW / rows;
  const int cells = tiles * cpg;
  // cell = (tile, channel): n = rows values with sum s_i and M2_i (squared deviations from the cell's own mean).  Pass 1:
  // the group mean; pass 2: M2 = sum_i [ M2_i + n (mean_i - mean)^2 ] — no difference of large numbers anywhere.
  double s = 0.0;
  for (int idx = lane; idx < cells; idx += 64) {
    const int ti = idx / cpg, c = g * cpg + (idx - ti * cpg);
    const long long tile = (long long)b * tiles + ti;
    s += c < N1 ? p1[tile * 2 * N1 + c] : p2[tile * 2 * N2 + (c - N1)];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const double n = (double)HW * cpg, mean = s / n;
  double q = 0.0;
  for (int idx = lane; idx < cells; idx += 64) {
    const int ti = idx / cpg, c = g * cpg + (idx - ti * cpg);
    const long long tile = (long long)b * tiles + ti;
    const double si = c < N1 ? p1[tile * 2 * N1 + c] : p2[tile * 2 * N2 + (c - N1)];
    const double qi = c < N1 ? p1[tile * 2 * N1 + N1 + c] : p2[tile * 2 * N2 + N2 + (c - N1)];
    const double dm = si / rows - mean;
    q += qi + dm * dm * rows;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  double var = q / n;
  var = var > 0.0 ? var : 0.0;
  if (mean_var && lane == 0) {
    mean_var[(long long)b * 2 * groups + g] = (float)mean;
    mean_var[(long long)b * 2 * groups + groups + g] = (float)var;
  }
  if (scale_shift) {
    const float rstd = rsqrtf((float)var + eps), mf = (float)mean;
    for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
      const float a = rstd * gamma[c];
      scale_shift[(long long)b * 2 * C + c] = a;
      scale_shift[(long long)b * 2 * C + C + c] = beta[c] - mf * a;
    }
  }
}
}  // namespace

namespace {
// gn_from_partials + gn_apply in ONE launch (round 5: 74 statistics-merge launches of ~5 us per network evaluation folded into
// their consumers).  Every block first merges the per (row tile, column) [sum, M2] cells of ITS sample for all groups — the same
// two-pass f64 merge as gn_from_partials (group mean, then M2 = sum_i [M2_i + n (mean_i - mean)^2]), TPG threads per group with
// a fixed-order combine through LDS — then streams y = act(x * a + s) exactly like gn_apply.  The cells of a sample are
// tiles * C f32 pairs (<= 40 KB, L2-resident: every block of the sample reads the same ones).
template <typename T>
__global__ void gn_apply_partials(const u16* __restrict__ x, long long ldx, u16* __restrict__ y, long long ldy,
                                  const float* __restrict__ p1, int N1, const float* __restrict__ p2, int N2, int rows,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int C, int groups,
                                  float eps, int rows_per_blk, int rpi, int silu) {
  __shared__ float stat[128];      // mean[groups], rstd[groups]
  __shared__ double red[1024];
  __shared__ double gmean[64];
  const int CV = C >> 3;
  const int t = threadIdx.x, nt = blockDim.x;
  const int b = blockIdx.y;
  const int cpg = C / groups, tiles = HW / rows, cells = tiles * cpg;
  const int TPG = nt / groups;                  // >= 2 (launcher)
  const int g = t / TPG, j = t - g * TPG;
  const bool act = g < groups;
  const double n = (double)HW * cpg;
  double s = 0.0;
  if (act) {
    for (int idx = j; idx < cells; idx += TPG) {
      const int ti = idx / cpg, c = g * cpg + (idx - ti * cpg);
      const long long tile = (long long)b * tiles + ti;
      s += c < N1 ? p1[tile * 2 * N1 + c] : p2[tile * 2 * N2 + (c - N1)];
    }
  }
  red[t] = s;
  __syncthreads();
  if (act && j == 0) {
    double a = 0.0;
    for (int k = 0; k < TPG; ++k) a += red[g * TPG + k];
    gmean[g] = a / n;
  }
  __syncthreads();
  double q = 0.0;
  if (act) {
    const double mean = gmean[g];
    for (int idx = j; idx < cells; idx += TPG) {
      const int ti = idx / cpg, c = g * cpg + (idx - ti * cpg);
      const long long tile = (long long)b * tiles + ti;
      const double si = c < N1 ? p1[tile * 2 * N1 + c] : p2[tile * 2 * N2 + (c - N1)];
      const double qi = c < N1 ? p1[tile * 2 * N1 + N1 + c] : p2[tile * 2 * N2 + N2 + (c - N1)];
      const double dm = si / rows - mean;
      q += qi + dm * dm * rows;
    }
  }
  red[t] = q;
  __syncthreads();
  if (act && j == 0) {
    double a = 0.0;
    for (int k = 0; k < TPG; ++k) a += red[g * TPG + k];
    double var = a / n;
    var = var > 0.0 ? var : 0.0;
    stat[g] = (float)gmean[g];
    stat[groups + g] = rsqrtf((float)var + eps);
  }
  __syncthreads();
  const int cv = t % CV, rsub = t / CV;
  float av[8], sv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cv * 8 + e;
    const int gg = c / cpg;
    const float a = stat[groups + gg] * gamma[c];
    av[e] = a;
    sv[e] = beta[c] - stat[gg] * a;
  }
  const int r_begin = blockIdx.x * rows_per_blk;
  const int r_end = min(HW, r_begin + rows_per_blk);
  const u16* xb = x + (long long)b * HW * ldx + cv * 8;
  u16* yb = y + (long long)b * HW * ldy + cv * 8;
  for (int r = r_begin + rsub; r < r_end; r += rpi) {
    const uint4 v = *reinterpret_cast<const uint4*>(xb + (long long)r * ldx);
    float f[8];
    unpack8<T>(v, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float o = f[e] * av[e] + sv[e];
      f[e] = silu ? silu_f(o) : o;
    }
    *reinterpret_cast<uint4*>(yb + (long long)r * ldy) = pack8<T>(f);
  }
}
}  // namespace

static int gn_geometry(int B, int HW, int C, int groups, int& nchunk, int& rows_per_chunk, int& rpi, int& threads);

extern "C" int dbir_groupnorm_apply_partials(int dtype, const void* x, long long ldx, void* y, long long ldy,
                                             const float* gamma, const float* beta, const float* p1, int N1, const float* p2,
                                             int N2, int rows, int B, int HW, int groups, float eps, int silu, void* stream) {
  const int C = N1 + N2;
  DBIR_CHECK_ARG(x && y && gamma && beta && p1 && N1 > 0 && (N2 == 0 || p2) && N2 >= 0,
                 "dbir_groupnorm_apply_partials: null pointer / bad producers");
  DBIR_CHECK_ARG(rows > 0 && HW % rows == 0 && B > 0 && B <= 65535 && groups > 0 && groups <= 64 && C % groups == 0 &&
                     C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && C <= 4096,
                 "dbir_groupnorm_apply_partials: need HW %% rows == 0 (HW %d, rows %d), groups <= 64 dividing C, C %% 8 == 0", HW,
                 rows);
  int nchunk, rows_per_chunk, rpi, threads;
  DBIR_CHECK_ARG(gn_geometry(B, HW, C, groups, nchunk, rows_per_chunk, rpi, threads),
                 "dbir_groupnorm_apply_partials: unsupported C / groups combination");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int nblk = (512 + B - 1) / B;   // fewer, fatter blocks than gn_apply: each one repeats the merge of its sample's cells
  int rows_per_blk = (HW + nblk - 1) / nblk;
  if (rows_per_blk < rpi) rows_per_blk = rpi;
  nblk = (HW + rows_per_blk - 1) / rows_per_blk;
  if (dtype == DBIR_F16)
    hipLaunchKernelGGL((gn_apply_partials<F16>), dim3(nblk, B), dim3(threads), 0, s, (const u16*)x, ldx, (u16*)y, ldy, p1, N1,
                       p2, N2, rows, gamma, beta, HW, C, groups, eps, rows_per_blk, rpi, silu);
  else if (dtype == DBIR_BF16)
    hipLaunchKernelGGL((gn_apply_partials<BF16>), dim3(nblk, B), dim3(threads), 0, s, (const u16*)x, ldx, (u16*)y, ldy, p1, N1,
                       p2, N2, rows, gamma, beta, HW, C, groups, eps, rows_per_blk, rpi, silu);
  else {
    dbir_set_error("dbir_groupnorm_apply_partials: bad dtype");
    return DBIR_ERR_ARG;
  }
  DBIR_CHECK_LAUNCH("dbir_groupnorm_apply_partials");
  return DBIR_OK;
}

extern "C" int dbir_groupnorm_from_partials(const float* p1, int N1, const float* p2, int N2, int rows, int B, int HW,
                                            int groups, float eps, const float* gamma, const float* beta, float* mean_var,
                                            float* scale_shift, void* stream) {
  DBIR_CHECK_ARG(p1 && N1 > 0 && (N2 == 0 || p2) && N2 >= 0, "dbir_groupnorm_from_partials: bad producers");
  DBIR_CHECK_ARG(rows > 0 && HW % rows == 0 && B > 0 && B <= 65535 && groups > 0 && groups <= 64 && (N1 + N2) % groups == 0,
                 "dbir_groupnorm_from_partials: need HW %% rows == 0 (HW %d, rows %d), groups <= 64 dividing C", HW, rows);
  DBIR_CHECK_ARG(mean_var || scale_shift, "dbir_groupnorm_from_partials: no output requested");
  DBIR_CHECK_ARG(!scale_shift || (gamma && beta), "dbir_groupnorm_from_partials: scale_shift needs gamma / beta");
  hipLaunchKernelGGL(gn_from_partials, dim3((groups + 3) / 4, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p1, N1,
                     p2, N2, rows, HW, groups, eps, gamma, beta, mean_var, scale_shift);
  DBIR_CHECK_LAUNCH("dbir_groupnorm_from_partials");
  return DBIR_OK;
}

extern "C" int dbir_groupnorm_nchunk(int HW, int C) {
  (void)C;
  int n = (HW + 15) / 16;
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}

extern "C" int dbir_groupnorm(int dtype, const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                              const float* beta, int B, int HW, int C, int groups, float eps, int silu,
                              float* workspace, void* stream) {
  DBIR_CHECK_ARG(x && y && gamma && beta && workspace, "dbir_groupnorm: null pointer");
  DBIR_CHECK_ARG(C % 8 == 0 && C % groups == 0 && ldx % 8 == 0 && ldy % 8 == 0 && C <= 4096,
                 "dbir_groupnorm: need C%%8==0, C%%groups==0, ld%%8==0 (C=%d)", C);
  DBIR_CHECK_ARG(groups <= 64 && B > 0 && B <= 65535, "dbir_groupnorm: groups must be <= 64, B <= 65535");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // ~512 blocks per launch (2 per CU), never more chunks than the workspace contract (dbir_groupnorm_nchunk)
  int nchunk = (512 + B - 1) / B;
  const int cap = dbir_groupnorm_nchunk(HW, C);
  if (nchunk > cap) nchunk = cap;
  const int rows_per_chunk = (HW + nchunk - 1) / nchunk;
  nchunk = (HW + rows_per_chunk - 1) / rows_per_chunk;
  const int CV = C / 8;
  const int rpi = CV >= 256 ? 1 : 256 / CV;
  const int threads = CV * rpi;
  DBIR_CHECK_ARG(threads <= 1024 && threads >= 2 * groups, "dbir_groupnorm: unsupported C / groups combination");
  float* partial = workspace;  // [B][nchunk][2*groups]
  const size_t sh1 = (size_t)rpi * 2 * C * sizeof(float);
  int nblk = (1024 + B - 1) / B;
  int rows_per_blk = (HW + nblk - 1) / nblk;
  if (rows_per_blk < rpi) rows_per_blk = rpi;
  nblk = (HW + rows_per_blk - 1) / rows_per_blk;
#define GN_LAUNCH(TT)                                                                                              \
  do {                                                                                                             \
    hipLaunchKernelGGL((gn_stats<TT>), dim3(nchunk, B), dim3(threads), sh1, s, (const u16*)x, ldx, partial, HW, C,  \
                       groups, rows_per_chunk, rpi);                                                               \
    hipLaunchKernelGGL((gn_apply<TT>), dim3(nblk, B), dim3(threads), 0, s, (const u16*)x, ldx, (u16*)y, ldy,       \
                       partial, gamma, beta, nchunk, HW, C, groups, eps, rows_per_blk, rpi, silu,                  \
                       (const float*)nullptr);                                                                     \
  } while (0)
  if (dtype == DBIR_F16)
    GN_LAUNCH(F16);
  else if (dtype == DBIR_BF16)
    GN_LAUNCH(BF16);
  else {
    dbir_set_error("dbir_groupnorm: bad dtype");
    return DBIR_ERR_ARG;
  }
#undef GN_LAUNCH
  DBIR_CHECK_LAUNCH("dbir_groupnorm");
  return DBIR_OK;
}

// Split form for the tiled VAE (reference utils/tilevae/tilevae.py:232-304): statistics of one tile, and normalisation of a
// tile with statistics aggregated by the caller over all tiles.
static int gn_geometry(int B, int HW, int C, int groups, int& nchunk, int& rows_per_chunk, int& rpi, int& threads) {
  nchunk = (512 + B - 1) / B;
  const int cap = dbir_groupnorm_nchunk(HW, C);
  if (nchunk > cap) nchunk = cap;
  rows_per_chunk = (HW + nchunk - 1) / nchunk;
  nchunk = (HW + rows_per_chunk - 1) / rows_per_chunk;
  const int CV = C / 8;
  rpi = CV >= 256 ? 1 : 256 / CV;
  threads = CV * rpi;
  return threads <= 1024 && threads >= 2 * groups;
}

extern "C" int dbir_groupnorm_stats(int dtype, const void* x, long long ldx, int B, int HW, int C, int groups,
                                    float* workspace, float* mean_var, void* stream) {
  DBIR_CHECK_ARG(x && workspace && mean_var, "dbir_groupnorm_stats: null pointer");
  DBIR_CHECK_ARG(C % 8 == 0 && C % groups == 0 && ldx % 8 == 0 && C <= 4096 && groups <= 64 && B > 0 && B <= 65535,
                 "dbir_groupnorm_stats: need C%%8==0, C%%groups==0, ld%%8==0, groups<=64 (C=%d)", C);
  int nchunk, rows_per_chunk, rpi, threads;
  DBIR_CHECK_ARG(gn_geometry(B, HW, C, groups, nchunk, rows_per_chunk, rpi, threads),
                 "dbir_groupnorm_stats: unsupported C / groups combination");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t sh1 = (size_t)rpi * 2 * C * sizeof(float);
  if (dtype == DBIR_F16) {
    hipLaunchKernelGGL((gn_stats<F16>), dim3(nchunk, B), dim3(threads), sh1, s, (const u16*)x, ldx, workspace, HW, C,
                       groups, rows_per_chunk, rpi);
    hipLaunchKernelGGL((gn_finalize<F16>), dim3(B), dim3(64), 0, s, (const u16*)x, ldx, workspace, mean_var, nchunk, HW,
                       C, groups);
  } else if (dtype == DBIR_BF16) {
    hipLaunchKernelGGL((gn_stats<BF16>), dim3(nchunk, B), dim3(threads), sh1, s, (const u16*)x, ldx, workspace, HW, C,
                       groups, rows_per_chunk, rpi);
    hipLaunchKernelGGL((gn_finalize<BF16>), dim3(B), dim3(64), 0, s, (const u16*)x, ldx, workspace, mean_var, nchunk, HW,
                       C, groups);
  } else {
    dbir_set_error("dbir_groupnorm_stats: bad dtype");
    return DBIR_ERR_ARG;
  }
  DBIR_CHECK_LAUNCH("dbir_groupnorm_stats");
  return DBIR_OK;
}

extern "C" int dbir_groupnorm_affine(int dtype, const void* x, long long ldx, int B, int HW, int C, int groups, float eps,
                                     const float* gamma, const float* beta, float* workspace, float* scale_shift,
                                     void* stream) {
  DBIR_CHECK_ARG(x && workspace && scale_shift && gamma && beta, "dbir_groupnorm_affine: null pointer");
  DBIR_CHECK_ARG(C % 8 == 0 && C % groups == 0 && ldx % 8 == 0 && C <= 4096 && groups <= 64 && B > 0 && B <= 65535,
                 "dbir_groupnorm_affine: need C%%8==0, C%%groups==0, ld%%8==0, groups<=64 (C=%d)", C);
  int nchunk, rows_per_chunk, rpi, threads;
  DBIR_CHECK_ARG(gn_geometry(B, HW, C, groups, nchunk, rows_per_chunk, rpi, threads),
                 "dbir_groupnorm_affine: unsupported C / groups combination");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t sh1 = (size_t)rpi * 2 * C * sizeof(float);
  if (dtype == DBIR_F16) {
    hipLaunchKernelGGL((gn_stats<F16>), dim3(nchunk, B), dim3(threads), sh1, s, (const u16*)x, ldx, workspace, HW, C,
                       groups, rows_per_chunk, rpi);
    hipLaunchKernelGGL((gn_finalize_affine<F16>), dim3(B), dim3(256), 0, s, (const u16*)x, ldx, workspace, gamma, beta,
                       scale_shift, nchunk, HW, C, groups, eps);
  } else if (dtype == DBIR_BF16) {
    hipLaunchKernelGGL((gn_stats<BF16>), dim3(nchunk, B), dim3(threads), sh1, s, (const u16*)x, ldx, workspace, HW, C,
                       groups, rows_per_chunk, rpi);
    hipLaunchKernelGGL((gn_finalize_affine<BF16>), dim3(B), dim3(256), 0, s, (const u16*)x, ldx, workspace, gamma, beta,
                       scale_shift, nchunk, HW, C, groups, eps);
  } else {
    dbir_set_error("dbir_groupnorm_affine: bad dtype");
    return DBIR_ERR_ARG;
  }
  DBIR_CHECK_LAUNCH("dbir_groupnorm_affine");
  return DBIR_OK;
}

extern "C" int dbir_groupnorm_apply(int dtype, const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                                    const float* beta, const float* mean_var, int B, int HW, int C, int groups, float eps,
                                    int silu, void* stream) {
  DBIR_CHECK_ARG(x && y && gamma && beta && mean_var, "dbir_groupnorm_apply: null pointer");
  DBIR_CHECK_ARG(C % 8 == 0 && C % groups == 0 && ldx % 8 == 0 && ldy % 8 == 0 && C <= 4096 && groups <= 64 && B > 0 &&
                     B <= 65535, "dbir_groupnorm_apply: need C%%8==0, C%%groups==0, ld%%8==0, groups<=64 (C=%d)", C);
  int nchunk, rows_per_chunk, rpi, threads;
  DBIR_CHECK_ARG(gn_geometry(B, HW, C, groups, nchunk, rows_per_chunk, rpi, threads),
                 "dbir_groupnorm_apply: unsupported C / groups combination");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int nblk = (1024 + B - 1) / B;
  int rows_per_blk = (HW + nblk - 1) / nblk;
  if (rows_per_blk < rpi) rows_per_blk = rpi;
  nblk = (HW + rows_per_blk - 1) / rows_per_blk;
  if (dtype == DBIR_F16)
    hipLaunchKernelGGL((gn_apply<F16>), dim3(nblk, B), dim3(threads), 0, s, (const u16*)x, ldx, (u16*)y, ldy,
                       (const float*)nullptr, gamma, beta, 0, HW, C, groups, eps, rows_per_blk, rpi, silu, mean_var);
  else if (dtype == DBIR_BF16)
    hipLaunchKernelGGL((gn_apply<BF16>), dim3(nblk, B), dim3(threads), 0, s, (const u16*)x, ldx, (u16*)y, ldy,
                       (const float*)nullptr, gamma, beta, 0, HW, C, groups, eps, rows_per_blk, rpi, silu, mean_var);
  else {
    dbir_set_error("dbir_groupnorm_apply: bad dtype");
    return DBIR_ERR_ARG;
  }
  DBIR_CHECK_LAUNCH("dbir_groupnorm_apply");
  return DBIR_OK;
}

extern "C" int dbir_layernorm(int dtype, const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                              const float* beta, int rows, int C, int Cpad, float eps, void* stream) {
  DBIR_CHECK_ARG(x && y && gamma && beta, "dbir_layernorm: null pointer");
  DBIR_CHECK_ARG(Cpad % 8 == 0 && Cpad >= C && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= Cpad && ldy >= Cpad,
                 "dbir_layernorm: bad C/Cpad/ld");
  DBIR_CHECK_ARG(Cpad <= 64 * 8 * 4, "dbir_layernorm: C up to 2048 supported (got %d)", Cpad);
  DBIR_CHECK_ARG(rows > 0, "dbir_layernorm: bad rows");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nv = Cpad / 8;
#define LN2_LAUNCH(TT, LPR, CPL)                                                                                   \
  hipLaunchKernelGGL((ln2_kernel<TT, LPR, CPL>), dim3((unsigned)((rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)))),   \
                     dim3(256), 0, s, (const u16*)x, ldx, (u16*)y, ldy, gamma, beta, rows, C, eps)
#define LN2_TRY(TT)                                                                                                \
  do {                                                                                                             \
    if (nv == 40) { LN2_LAUNCH(TT, 8, 5); goto ln_done; }   /* C = 320 */                                          \
    if (nv == 80) { LN2_LAUNCH(TT, 16, 5); goto ln_done; }  /* C = 640 */                                          \
    if (nv == 160) { LN2_LAUNCH(TT, 32, 5); goto ln_done; } /* C = 1280 */                                         \
    if (nv == 24) { LN2_LAUNCH(TT, 8, 3); goto ln_done; }   /* SwinIR C = 180 (padded to 192) */                   \
    if (nv == 8) { LN2_LAUNCH(TT, 8, 1); goto ln_done; }                                                           \
  } while (0)
  if (dtype == DBIR_F16) LN2_TRY(F16);
  else if (dtype == DBIR_BF16) LN2_TRY(BF16);
#define LN_LAUNCH(TT, MV, RW)                                                                                     \
  hipLaunchKernelGGL((ln_kernel<TT, MV, RW>), dim3((rows + 4 * RW - 1) / (4 * RW)), dim3(256), 0, s, (const u16*)x, \
                     ldx, (u16*)y, ldy, gamma, beta, rows, C, Cpad, eps)
  if (dtype == DBIR_F16) {
    if (nv <= 64) LN_LAUNCH(F16, 1, 4);
    else if (nv <= 128) LN_LAUNCH(F16, 2, 2);
    else LN_LAUNCH(F16, 4, 1);
  } else if (dtype == DBIR_BF16) {
    if (nv <= 64) LN_LAUNCH(BF16, 1, 4);
    else if (nv <= 128) LN_LAUNCH(BF16, 2, 2);
    else LN_LAUNCH(BF16, 4, 1);
  } else {
    dbir_set_error("dbir_layernorm: bad dtype");
    return DBIR_ERR_ARG;
  }
#undef LN_LAUNCH
ln_done:
#undef LN2_TRY
#undef LN2_LAUNCH
  DBIR_CHECK_LAUNCH("dbir_layernorm");
  return DBIR_OK;
}

extern "C" int dbir_softmax_rows(int dtype, void* x, long long ld, long long rows, int L, void* stream) {
  DBIR_CHECK_ARG(x && ld % 8 == 0 && L > 0 && L <= ld, "dbir_softmax_rows: bad args");
  DBIR_CHECK_ARG(rows > 0 && rows < 2147483647LL, "dbir_softmax_rows: bad rows");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DBIR_F16)
    hipLaunchKernelGGL((softmax_kernel<F16>), dim3((unsigned)rows), dim3(256), 0, s, (u16*)x, ld, L);
  else if (dtype == DBIR_BF16)
    hipLaunchKernelGGL((softmax_kernel<BF16>), dim3((unsigned)rows), dim3(256), 0, s, (u16*)x, ld, L);
  else {
    dbir_set_error("dbir_softmax_rows: bad dtype");
    return DBIR_ERR_ARG;
  }
  DBIR_CHECK_LAUNCH("dbir_softmax_rows");
  return DBIR_OK;
}
