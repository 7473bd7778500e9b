// GroupNorm(+SiLU), LayerNorm and row softmax for NHWC 16-bit activations (HBM-bound kernels, f32 statistics).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// GroupNorm, three launches:
//  1. gn_partial : grid (nchunk, B). Thread owns one 8-channel vector (16 B loads, coalesced along C) and walks
//                  the chunk's rows; per-channel (sum, sumsq) combined through LDS in a fixed order
//                  -> partial[b][chunk][2C] (deterministic: no float atomics anywhere).
//  2. gn_finalize: grid (B). Sums the chunks, reduces channels -> groups, emits per-(b,c) affine coefficients
//                  a = rstd*gamma, s = beta - mean*rstd*gamma.
//  3. gn_apply   : y = act(x*a + s), vectorised elementwise.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void gn_partial(const u16* __restrict__ x, long long ldx, float* __restrict__ partial, int HW, int C,
                           int rows_per_chunk, int rpi) {
  extern __shared__ float sh[];  // [rpi][2*C]: per row-subset partial sums, combined in a FIXED order below
  const int CV = C >> 3;
  const int t = threadIdx.x;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cv = t % CV, rsub = t / CV;
  const int r_begin = chunk * rows_per_chunk;
  const int r_end = min(HW, r_begin + rows_per_chunk);
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
  const u16* xb = x + (long long)b * HW * ldx + cv * 8;
  for (int r = r_begin + rsub; r < r_end; r += rpi) {
    const uint4 v = *reinterpret_cast<const uint4*>(xb + (long long)r * ldx);
    float f[8];
    unpack8<T>(v, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] += f[e];
      ss[e] += f[e] * f[e];
    }
  }
  float* mine = sh + (long long)rsub * 2 * C;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mine[cv * 8 + e] = s[e];
    mine[C + cv * 8 + e] = ss[e];
  }
  __syncthreads();
  // deterministic combine (no float atomics: results are bit-reproducible run to run and batch-size independent)
  float* out = partial + ((long long)b * gridDim.x + chunk) * 2 * C;
  for (int i = t; i < 2 * C; i += blockDim.x) {
    float a = 0.f;
    for (int k = 0; k < rpi; ++k) a += sh[(long long)k * 2 * C + i];
    out[i] = a;
  }
}

__global__ void gn_finalize(const float* __restrict__ partial, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float* __restrict__ coef, int nchunk, int HW, int C,
                            int groups, float eps) {
  extern __shared__ float sh[];  // [2*C] per-channel sums, then [2*groups] mean / rstd
  const int b = blockIdx.x, t = threadIdx.x;
  const int cpg = C / groups;
  const float* pb = partial + (long long)b * nchunk * 2 * C;
  for (int c = t; c < 2 * C; c += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += pb[(long long)k * 2 * C + c];
    sh[c] = s;
  }
  __syncthreads();
  float* stat = sh + 2 * C;
  if (t < groups) {
    float s = 0.f, ss = 0.f;
    for (int c = t * cpg; c < (t + 1) * cpg; ++c) {
      s += sh[c];
      ss += sh[C + c];
    }
    const float n = (float)HW * (float)cpg;
    const float mean = s / n;
    const float var = fmaxf(ss / n - mean * mean, 0.f);
    stat[t] = mean;
    stat[groups + t] = rsqrtf(var + eps);
  }
  __syncthreads();
  float* cb = coef + (long long)b * 2 * C;
  for (int c = t; c < C; c += blockDim.x) {
    const int g = c / cpg;
    const float a = stat[groups + g] * gamma[c];
    cb[c] = a;
    cb[C + c] = beta[c] - stat[g] * a;
  }
}

template <typename T>
__global__ void gn_apply(const u16* __restrict__ x, long long ldx, u16* __restrict__ y, long long ldy,
                         const float* __restrict__ coef, int HW, int C, long long total_vec, int silu) {
  const int CV = C >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / CV;
    const int cv = (int)(i - row * CV);
    const int b = (int)(row / HW);
    const uint4 v = *reinterpret_cast<const uint4*>(x + row * ldx + cv * 8);
    float f[8];
    unpack8<T>(v, f);
    const float* ca = coef + (long long)b * 2 * C + cv * 8;
    const float4 a0 = *reinterpret_cast<const float4*>(ca), a1 = *reinterpret_cast<const float4*>(ca + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(ca + C), s1 = *reinterpret_cast<const float4*>(ca + C + 4);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float o = f[e] * av[e] + sv[e];
      f[e] = silu ? silu_f(o) : o;
    }
    *reinterpret_cast<uint4*>(y + row * ldy + cv * 8) = pack8<T>(f);
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave64 per row, row held in registers (two-pass statistics, f32).
// ------------------------------------------------------------------------------------------------
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void ln_kernel(const u16* __restrict__ x, long long ldx, u16* __restrict__ y,
                                                 long long ldy, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, int rows, int C, int Cpad,
                                                 float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int NV = Cpad >> 3;
  float f[MAXV][8];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int v = lane + 64 * k;
    if (v < NV) {
      const uint4 q = *reinterpret_cast<const uint4*>(x + row * ldx + v * 8);
      unpack8<T>(q, f[k]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (v * 8 + e >= C) f[k][e] = 0.f;
        sum += f[k][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[k][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int v = lane + 64 * k;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dlt = (v * 8 + e < C) ? f[k][e] - mean : 0.f;
      sq += dlt * dlt;
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int v = lane + 64 * k;
    if (v < NV) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = v * 8 + e;
        o[e] = c < C ? (f[k][e] - mean) * rstd * gamma[c] + beta[c] : 0.f;
      }
      *reinterpret_cast<uint4*>(y + row * ldy + v * 8) = pack8<T>(o);
    }
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
  DBIR_CHECK_ARG(groups <= 64, "dbir_groupnorm: groups must be <= 64");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nchunk = dbir_groupnorm_nchunk(HW, C);
  const int rows_per_chunk = (HW + nchunk - 1) / nchunk;
  const int CV = C / 8;
  const int rpi = CV >= 256 ? 1 : 256 / CV;
  const int threads = CV * rpi;
  DBIR_CHECK_ARG(threads <= 1024, "dbir_groupnorm: C too large");
  float* partial = workspace;
  float* coef = workspace + (long long)B * nchunk * 2 * C;
  const size_t sh1 = (size_t)rpi * 2 * C * sizeof(float);
  const long long total_vec = (long long)B * HW * CV;
  const int ablocks = (int)((total_vec + 255) / 256 > 4096 ? 4096 : (total_vec + 255) / 256);
  if (dtype == DBIR_F16) {
    hipLaunchKernelGGL((gn_partial<F16>), dim3(nchunk, B), dim3(threads), sh1, s, (const u16*)x, ldx, partial, HW, C,
                       rows_per_chunk, rpi);
  } else if (dtype == DBIR_BF16) {
    hipLaunchKernelGGL((gn_partial<BF16>), dim3(nchunk, B), dim3(threads), sh1, s, (const u16*)x, ldx, partial, HW,
                       C, rows_per_chunk, rpi);
  } else {
    dbir_set_error("dbir_groupnorm: bad dtype");
    return DBIR_ERR_ARG;
  }
  hipLaunchKernelGGL(gn_finalize, dim3(B), dim3(256), (size_t)(2 * C + 2 * groups) * sizeof(float), s, partial, gamma, beta, coef,
                     nchunk, HW, C, groups, eps);
  if (dtype == DBIR_F16)
    hipLaunchKernelGGL((gn_apply<F16>), dim3(ablocks), dim3(256), 0, s, (const u16*)x, ldx, (u16*)y, ldy, coef, HW,
                       C, total_vec, silu);
  else
    hipLaunchKernelGGL((gn_apply<BF16>), dim3(ablocks), dim3(256), 0, s, (const u16*)x, ldx, (u16*)y, ldy, coef,
                       HW, C, total_vec, silu);
  DBIR_CHECK_LAUNCH("dbir_groupnorm");
  return DBIR_OK;
}

extern "C" int dbir_layernorm(int dtype, const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                              const float* beta, int rows, int C, int Cpad, float eps, void* stream) {
  DBIR_CHECK_ARG(x && y && gamma && beta, "dbir_layernorm: null pointer");
  DBIR_CHECK_ARG(Cpad % 8 == 0 && Cpad >= C && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= Cpad && ldy >= Cpad,
                 "dbir_layernorm: bad C/Cpad/ld");
  DBIR_CHECK_ARG(Cpad <= 64 * 8 * 4, "dbir_layernorm: C up to 2048 supported (got %d)", Cpad);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int blocks = (rows + 3) / 4;
  const int nv = Cpad / 8;
#define LN_LAUNCH(TT, MV)                                                                                         \
  hipLaunchKernelGGL((ln_kernel<TT, MV>), dim3(blocks), dim3(256), 0, s, (const u16*)x, ldx, (u16*)y, ldy, gamma, \
                     beta, rows, C, Cpad, eps)
  if (dtype == DBIR_F16) {
    if (nv <= 64) LN_LAUNCH(F16, 1);
    else if (nv <= 128) LN_LAUNCH(F16, 2);
    else LN_LAUNCH(F16, 4);
  } else if (dtype == DBIR_BF16) {
    if (nv <= 64) LN_LAUNCH(BF16, 1);
    else if (nv <= 128) LN_LAUNCH(BF16, 2);
    else LN_LAUNCH(BF16, 4);
  } else {
    dbir_set_error("dbir_layernorm: bad dtype");
    return DBIR_ERR_ARG;
  }
#undef LN_LAUNCH
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
