// Shared epilogue of the direct-to-LDS implicit-GEMM kernels (gemm_glds.hip, gemm_halo.hip): split-K slab store, or
// f32 math in registers (bias, time-embedding row vector, SiLU / GELU / LeakyReLU / GEGLU, scale) -> 16-bit tile
// transposed through LDS -> row-contiguous 16-byte stores (+ residual), or the per-batch transposed (V^T) store.
// The accumulators are D[n][m] (MFMA first operand = weight rows): a lane holds 4 consecutive output channels of one
// pixel.  `smem` is the block's whole dynamic LDS (the operand ring is dead when this runs: the caller's main loop has
// retired every direct-to-LDS load; the function starts with the barrier that orders the last fragment reads).
#pragma once
#include "common.h"

struct EpiParams {
  int vec_bias, vec_rv;  // bias / row vector may be read with 16 B / 8 B vector loads
  int splitk;            // number of K slices (1 = off)
  float* ws;             // split-K: f32 partial sums [z][slice][M][N]
};

// exact-GELU with erf from Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the 16-bit output ulp):
// ~12 VALU + 1 exp + 1 rcp instead of the branchy library erff.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float e = 1.0f - poly * t * __expf(-z * z);  // erf(|x|/sqrt2)
  const float erfv = x < 0.f ? -e : e;
  return 0.5f * x * (1.0f + erfv);
}

// ---- GroupNorm column statistics of the STORED tile (dbir_gemm_desc.stats) -------------------------------------------------
// Per (row tile, column): stats[tile][0][n] = sum of the stored values, stats[tile][1][n] = M2 = sum of squared deviations
// from that tile-column's own mean.  A thread accumulates its rows SHIFTED by its first value (no cancellation when
// |mean| >> sigma: ADVICE round 3), the row lanes of a column are merged pairwise (Chan et al.) in a fixed order through
// LDS: deterministic, and the consumer (gn_from_partials) merges tiles the same way in f64.
struct ColStat {
  float piv[8], s1[8], s2[8];
  int cnt;
};
__device__ __forceinline__ void colstat_init(ColStat& c) {
#pragma unroll
  for (int e = 0; e < 8; ++e) c.piv[e] = c.s1[e] = c.s2[e] = 0.f;
  c.cnt = 0;
}
__device__ __forceinline__ void colstat_add(ColStat& c, const float* a) {
  if (c.cnt == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) c.piv[e] = a[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float dlt = a[e] - c.piv[e];
    c.s1[e] += dlt;
    c.s2[e] += dlt * dlt;
  }
  ++c.cnt;
}
// CPR chunks of 8 columns per tile row, RL row lanes (thread (ch, rl) walked rows rl, rl + RL, ...); `active` = the thread
// took part; smem: the block's LDS (>= RL * CPR * 68 bytes, nothing else live); NT threads call this together.
template <int NT, int BN, int CPR, int RL>
__device__ __forceinline__ void colstat_finish(const ColStat& c, bool active, char* smem, int tid, int ch, int rl,
                                               float* __restrict__ st, int tn, int N) {
  __syncthreads();
  float* ps = reinterpret_cast<float*>(smem);          // [RL * CPR][16]: mean[8] | M2[8]
  int* pc = reinterpret_cast<int*>(smem + RL * CPR * 64);  // [RL * CPR]: rows accumulated
  if (rl < RL) {
    const int slot = rl * CPR + ch;
    const int n = active ? c.cnt : 0;
    const float inv = n > 0 ? 1.0f / (float)n : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ps[slot * 16 + e] = c.piv[e] + c.s1[e] * inv;
      ps[slot * 16 + 8 + e] = c.s2[e] - c.s1[e] * c.s1[e] * inv;
    }
    pc[slot] = n;
  }
  __syncthreads();
  for (int col = tid; col < BN; col += NT) {
    const int n = tn * BN + col;
    if (n >= N) continue;
    // two passes over the row lanes, ONE division: mean = sum n_r mean_r / sum n_r, then M2 = sum [M2_r + n_r (mean_r - mean)^2]
    // (the lanes' means are close to each other: no difference of large numbers; fixed order: deterministic)
    float cnt = 0.f, sum = 0.f;
    for (int r = 0; r < RL; ++r) {
      const int slot = r * CPR + (col >> 3);
      const float nr = (float)pc[slot];
      cnt += nr;
      sum += nr * ps[slot * 16 + (col & 7)];
    }
    const float mean = cnt > 0.f ? sum / cnt : 0.f;
    float m2 = 0.f;
    for (int r = 0; r < RL; ++r) {
      const int slot = r * CPR + (col >> 3);
      const float nr = (float)pc[slot], dlt = ps[slot * 16 + (col & 7)] - mean;
      m2 += ps[slot * 16 + 8 + (col & 7)] + nr * dlt * dlt;
    }
    st[n] = sum;
    st[N + n] = m2 > 0.f ? m2 : 0.f;
  }
}

template <typename T, int WM, int WN, int MI, int NJ>
__device__ __forceinline__ void gemm_epilogue(const dbir_gemm_desc& d, const EpiParams& p, f32x16 (&acc)[MI][NJ],
                                              char* smem, int tm, int tn, int ksp, int bz, int wm, int wn, int tid) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
  const int lane = tid & 63, lq = lane & 31, hi = lane >> 5;
  const int M = d.M;
  const int N = d.N;
  if (p.splitk > 1) {
    // split-K: raw f32 partial sums to the workspace; bias / epilogue / 16-bit store happen in splitk_reduce_kernel
    float* __restrict__ wsp = p.ws + ((long long)bz * p.splitk + ksp) * (long long)M * N;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = tm * BM + wm * 32 * MI + i * 32 + lq;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n0 = tn * BN + wn * 32 * NJ + j * 32 + 8 * g + 4 * hi;
          if (n0 + 4 <= N) {  // N % 8 == 0 is required for split-K, so 16-byte aligned
            *reinterpret_cast<float4*>(wsp + (long long)m * N + n0) =
                make_float4(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          }
        }
    }
    return;
  }
  if (d.out_f32) {
    // f32 output (network heads with a handful of channels: UNet `out` 320 -> 4, unet.py:678): each lane stores its 4
    // consecutive channels straight from registers as one float4 (ldc % 4 == 0, no residual / GEGLU / transposed store)
    float* __restrict__ Cf = reinterpret_cast<float*>(d.C) + (long long)bz * d.strideC_z;
    const u16* __restrict__ RVf = reinterpret_cast<const u16*>(d.rowvec);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = tm * BM + wm * 32 * MI + i * 32 + lq;
      if (m >= M) continue;
      const u16* rvp = RVf ? RVf + (long long)(m / d.rows_per_batch) * d.rowvec_ld : nullptr;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n0 = tn * BN + wn * 32 * NJ + j * 32 + 8 * g + 4 * hi;
          if (n0 >= N) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[i][j][4 * g + e];
            if (d.bias && n0 + e < N) x += d.bias[n0 + e];
            if (rvp && n0 + e < N) x += T::to_f32(rvp[n0 + e]);
            if (d.act == DBIR_ACT_SILU) x = silu_f(x);
            else if (d.act == DBIR_ACT_GELU) x = gelu_fast(x);
            else if (d.act == DBIR_ACT_LRELU) x = x > 0.f ? x : x * d.act_param;
            v[e] = x * d.out_scale;
          }
          float* dst = Cf + (long long)m * d.ldc + n0;
          if (n0 + 4 <= N) {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n0 + e < N) dst[e] = v[e];
          }
        }
    }
    return;
  }
  // ---------------- epilogue: f32 math in registers -> 16-bit tile in LDS -> row-contiguous 16 B stores ----------
  const bool geglu = d.act == DBIR_ACT_GEGLU;
  const int bn_out = geglu ? BN / 2 : BN;
  const int cs_ld = bn_out + 8;  // halfs; (bn_out + 8) * 2 B is a multiple of 16
  u16* Cs = reinterpret_cast<u16*>(smem);
  const u16* __restrict__ RV = reinterpret_cast<const u16*>(d.rowvec);

  // bias for this lane's columns (independent of the row tile): [j][g] -> 4 consecutive columns
  float4 b4[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n0 = tn * BN + wn * 32 * NJ + j * 32 + 8 * g + 4 * hi;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d.bias) {
        if (p.vec_bias && n0 + 4 <= N) {
          b = *reinterpret_cast<const float4*>(d.bias + n0);
        } else {
          if (n0 + 0 < N) b.x = d.bias[n0 + 0];
          if (n0 + 1 < N) b.y = d.bias[n0 + 1];
          if (n0 + 2 < N) b.z = d.bias[n0 + 2];
          if (n0 + 3 < N) b.w = d.bias[n0 + 3];
        }
      }
      b4[j][g] = b;
    }

  __syncthreads();  // all waves finished reading the operand tiles (no glds in flight any more)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wm * 32 * MI + i * 32 + lq;
    const int m = tm * BM + row;
    const int mb = (m < M ? m : M - 1);
    const u16* rvp = RV ? RV + (long long)(mb / d.rows_per_batch) * d.rowvec_ld : nullptr;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (geglu && (j & 1)) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = wn * 32 * NJ + j * 32 + 8 * g + 4 * hi;  // local packed column of element 0
        const int n0 = tn * BN + nl;
        float v[4] = {acc[i][j][4 * g + 0] + b4[j][g].x, acc[i][j][4 * g + 1] + b4[j][g].y,
                      acc[i][j][4 * g + 2] + b4[j][g].z, acc[i][j][4 * g + 3] + b4[j][g].w};
        if (rvp) {
          if (p.vec_rv && n0 + 4 <= N) {
            const uint2 rr = *reinterpret_cast<const uint2*>(rvp + n0);
            v[0] += T::to_f32((u16)(rr.x & 0xffff));
            v[1] += T::to_f32((u16)(rr.x >> 16));
            v[2] += T::to_f32((u16)(rr.y & 0xffff));
            v[3] += T::to_f32((u16)(rr.y >> 16));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n0 + e < N) v[e] += T::to_f32(rvp[n0 + e]);
          }
        }
        if (d.act == DBIR_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
        } else if (d.act == DBIR_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
        } else if (d.act == DBIR_ACT_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * d.act_param;
        } else if (geglu) {
          if constexpr (NJ == 2) {
            constexpr int JG = 1;  // gate tile of this value tile (NJ == 2: tiles are (value, gate))
            const float gt[4] = {acc[i][JG][4 * g + 0] + b4[JG][g].x, acc[i][JG][4 * g + 1] + b4[JG][g].y,
                                 acc[i][JG][4 * g + 2] + b4[JG][g].z, acc[i][JG][4 * g + 3] + b4[JG][g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= gelu_fast(gt[e]);
          }
        }
        const int ocl = geglu ? (wn * 16 * NJ + 8 * g + 4 * hi) : nl;
        uint2 pk;
        pk.x = (uint32_t)T::from_f32(v[0] * d.out_scale) | ((uint32_t)T::from_f32(v[1] * d.out_scale) << 16);
        pk.y = (uint32_t)T::from_f32(v[2] * d.out_scale) | ((uint32_t)T::from_f32(v[3] * d.out_scale) << 16);
        *reinterpret_cast<uint2*>(Cs + row * cs_ld + ocl) = pk;
      }
    }
  }
  __syncthreads();
  if (d.store_mode == 1) {
    // transposed per-batch store  C[(m / L) * bstride + n * trans_ld + (m % L)]  (V^T for the attention kernel):
    // one thread = 8 consecutive rows m of one column n -> one 16-byte store; consecutive lanes take consecutive
    // columns, so the 2-byte LDS reads are conflict-free (L % 8 == 0 keeps a chunk inside one batch).
    u16* __restrict__ Ct = reinterpret_cast<u16*>(d.C);
    const int total = (BM / 8) * bn_out;
    for (int q = tid; q < total; q += NT) {
      const int mc = q / bn_out, nl = q - mc * bn_out;
      const int m0 = tm * BM + mc * 8, n = tn * bn_out + nl;
      if (m0 >= M || n >= N) continue;
      u16 hv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) hv[e] = Cs[(mc * 8 + e) * cs_ld + nl];
      const int bb = m0 / d.trans_L, l0 = m0 - bb * d.trans_L;
      u16* dst = Ct + (long long)bb * d.trans_bstride + (long long)n * d.trans_ld + l0;
      if (m0 + 8 <= M) {
        uint4 v;
        v.x = (uint32_t)hv[0] | ((uint32_t)hv[1] << 16);
        v.y = (uint32_t)hv[2] | ((uint32_t)hv[3] << 16);
        v.z = (uint32_t)hv[4] | ((uint32_t)hv[5] << 16);
        v.w = (uint32_t)hv[6] | ((uint32_t)hv[7] << 16);
        *reinterpret_cast<uint4*>(dst) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (m0 + e < M) dst[e] = hv[e];
      }
    }
    return;
  }
  if (d.stats) {
    // ---- store + GroupNorm column sums of the STORED values (host side: no GEGLU, M % BM == 0, one z slice) --------------
    // a thread keeps ONE 8-column chunk and walks the tile's rows with stride RL, so its 8 sums / 8 sums of squares stay
    // in registers; the RL row lanes are then combined through LDS in a fixed order (deterministic) and every column of
    // the tile gets its two numbers: stats[tm][0][n] = sum, stats[tm][1][n] = sum of squares over the tile's BM rows.
    constexpr int CPR = BN / 8, RL = NT / CPR;
    const int ch = tid % CPR, rl = tid / CPR;
    const int ncol = tn * BN + ch * 8;
    const u16* __restrict__ Rg = d.R ? reinterpret_cast<const u16*>(d.R) : nullptr;
    u16* __restrict__ Cg = reinterpret_cast<u16*>(d.C);
    ColStat cs;
    colstat_init(cs);
    const bool active = rl < RL && ncol < N;
    if (active) {
      for (int row = rl; row < BM; row += RL) {
        const int m = tm * BM + row;
        float a[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(Cs + row * cs_ld + ch * 8), a);
        if (ncol + 8 <= N) {
          if (Rg) {
            float b[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(Rg + (long long)m * d.ldr + ncol), b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];
          }
          const uint4 v = pack8<T>(a);
          *reinterpret_cast<uint4*>(Cg + (long long)m * d.ldc + ncol) = v;
          unpack8<T>(v, a);  // statistics of what was stored (16-bit rounded)
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (ncol + e < N) {
              float x = a[e];
              if (Rg) x += T::to_f32(Rg[(long long)m * d.ldr + ncol + e]);
              const u16 hv = T::from_f32(x);
              Cg[(long long)m * d.ldc + ncol + e] = hv;
              a[e] = T::to_f32(hv);
            }
          }
        }
        colstat_add(cs, a);
      }
    }
    // [sum, M2] per column of the tile: shifted per-thread sums merged pairwise through LDS in a fixed order (round 4)
    colstat_finish<NT, BN, CPR, RL>(cs, active, smem, tid, ch, rl, d.stats + (long long)tm * 2 * N, tn, N);
  } else {
    const int n_out = geglu ? N / 2 : N;
    const int ch_per_row = bn_out >> 3;
    const int total = BM * ch_per_row;
    const u16* __restrict__ Rg = d.R ? reinterpret_cast<const u16*>(d.R) + (long long)bz * d.strideR_z : nullptr;
    u16* __restrict__ Cg = reinterpret_cast<u16*>(d.C) + (long long)bz * d.strideC_z;
    for (int q = tid; q < total; q += NT) {
      const int row = q / ch_per_row, ch = q - row * ch_per_row;
      const int m = tm * BM + row;
      const int ncol = tn * bn_out + ch * 8;
      if (m >= M || ncol >= n_out) continue;
      uint4 v = *reinterpret_cast<const uint4*>(Cs + row * cs_ld + ch * 8);
      if (ncol + 8 <= n_out) {
        if (Rg) {
          const uint4 rr = *reinterpret_cast<const uint4*>(Rg + (long long)m * d.ldr + ncol);
          float a[8], b[8];
          unpack8<T>(v, a);
          unpack8<T>(rr, b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
          v = pack8<T>(a);
        }
        *reinterpret_cast<uint4*>(Cg + (long long)m * d.ldc + ncol) = v;
      } else {
        float a[8];
        unpack8<T>(v, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (ncol + e < n_out) {
            float x = a[e];
            if (Rg) x += T::to_f32(Rg[(long long)m * d.ldr + ncol + e]);
            Cg[(long long)m * d.ldc + ncol + e] = T::from_f32(x);
          }
        }
      }
    }
  }
}
