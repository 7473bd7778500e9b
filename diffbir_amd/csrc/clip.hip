// OpenCLIP text tower glue kernels (reference diffbir/model/clip.py:37-54, open_clip/transformer.py ResidualAttentionBlock /
// text Transformer with the causal mask built at open_clip/model.py build_attention_mask).
//
// The tower is 23 pre-LN blocks over [B, 77, 1024]: its GEMMs (in_proj / out_proj / c_fc+GELU / c_proj) run on the
// engine's MFMA GEMM (dbir_gemm, 16-bit operands, f32 accumulation — what the reference's fp16 autocast does to
// F.linear), and this file supplies the three pieces around them:
//   * token + positional embedding gather -> the f32 residual stream;
//   * residual add + LayerNorm in one pass: x (f32, in place) += y (f32 GEMM output), LN(x) -> 16-bit operand of the
//     next GEMM (or f32 for ln_final).  The residual stream stays f32 like the reference's (autocast keeps LayerNorm
//     and the `x + attn(...)` sums in f32), so 23 blocks do not accumulate 16-bit rounding;
//   * causal multi-head attention for short sequences (L <= 128, head_dim 64): one workgroup per (batch, head), K / V
//     of the head in LDS as f32 (broadcast reads), one query row per thread, online softmax in f32.
// The whole tower is < 0.1 % of a restoration; the kernels are written for clarity and exactness, not for the roofline.
#include <stdint.h>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void clip_embed_kernel(const long long* __restrict__ tokens,
                                                         const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos, float* __restrict__ x, int L,
                                                         int W, int vocab) {
  const int row = blockIdx.x;  // b * L + i
  const int i = row % L;
  long long t = tokens[row];
  t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
  const float4* e = reinterpret_cast<const float4*>(tok_emb + t * W);
  const float4* p = reinterpret_cast<const float4*>(pos + (long long)i * W);
  float4* o = reinterpret_cast<float4*>(x + (long long)row * W);
  for (int c = threadIdx.x; c < W / 4; c += blockDim.x) {
    const float4 a = e[c], b = p[c];
    o[c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// one wave per row; 4 rows per workgroup
template <typename T>
__global__ __launch_bounds__(256) void add_ln_f32_kernel(float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, void* __restrict__ out,
                                                         long long ldo, int out_f32, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4* xr = reinterpret_cast<float4*>(x + (long long)row * C);
  const float4* yr = y ? reinterpret_cast<const float4*>(y + (long long)row * C) : nullptr;
  const int nv = C / 4;
  float sum = 0.f;
  for (int c = lane; c < nv; c += 64) {
    float4 v = xr[c];
    if (yr) {
      const float4 w = yr[c];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      xr[c] = v;
    }
    sum += (v.x + v.y) + (v.z + v.w);
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
  for (int c = lane; c < nv; c += 64) {  // a lane re-reads exactly what it wrote above
    const float4 v = xr[c];
    const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
    sq += (a * a + b * b) + (cc * cc + d * d);
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  for (int c = lane; c < nv; c += 64) {
    const float4 v = xr[c];
    const float4 g = reinterpret_cast<const float4*>(gamma)[c], b = reinterpret_cast<const float4*>(beta)[c];
    const float o0 = (v.x - mean) * rstd * g.x + b.x, o1 = (v.y - mean) * rstd * g.y + b.y;
    const float o2 = (v.z - mean) * rstd * g.z + b.z, o3 = (v.w - mean) * rstd * g.w + b.w;
    if (out_f32) {
      reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (long long)row * ldo)[c] = make_float4(o0, o1, o2, o3);
    } else {
      uint2 pk;
      pk.x = T::pack2(o0, o1);
      pk.y = T::pack2(o2, o3);
      reinterpret_cast<uint2*>(reinterpret_cast<u16*>(out) + (long long)row * ldo)[c] = pk;
    }
  }
}

constexpr int CA_MAXL = 128, CA_D = 64;

template <typename T>
__global__ __launch_bounds__(CA_MAXL) void causal_attn_kernel(const u16* __restrict__ qkv, long long ld,
                                                              u16* __restrict__ out, long long ldo, int H, int L,
                                                              float scale) {
  __shared__ float Ks[CA_MAXL][CA_D];
  __shared__ float Vs[CA_MAXL][CA_D];
  const int h = blockIdx.x, b = blockIdx.y, i = threadIdx.x;
  const long long HD = (long long)H * CA_D;
  const u16* base = qkv + (long long)b * L * ld + (long long)h * CA_D;
  // stage K and V of this head: thread -> (row, 8-column chunk)
  for (int q = i; q < L * (CA_D / 8); q += CA_MAXL) {
    const int r = q / (CA_D / 8), c = (q % (CA_D / 8)) * 8;
    float f[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(base + (long long)r * ld + HD + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) Ks[r][c + e] = f[e];
    unpack8<T>(*reinterpret_cast<const uint4*>(base + (long long)r * ld + 2 * HD + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) Vs[r][c + e] = f[e];
  }
  __syncthreads();
  const bool act = i < L;
  const int ii = act ? i : 0;
  float qv[CA_D], o[CA_D];
#pragma unroll
  for (int c = 0; c < CA_D; c += 8) {
    float f[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(base + (long long)ii * ld + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qv[c + e] = f[e] * scale;
      o[c + e] = 0.f;
    }
  }
  // the query at position i sees keys 0..i (additive -inf mask above the diagonal, open_clip build_attention_mask)
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j <= ii; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < CA_D; ++d) s += qv[d] * Ks[j][d];
    const float mn = fmaxf(m, s);
    const float alpha = __expf(m - mn), p = __expf(s - mn);
    l = l * alpha + p;
#pragma unroll
    for (int d = 0; d < CA_D; ++d) o[d] = o[d] * alpha + p * Vs[j][d];
    m = mn;
  }
  if (!act) return;
  const float inv = 1.f / l;
  u16* dst = out + ((long long)b * L + i) * ldo + (long long)h * CA_D;
#pragma unroll
  for (int c = 0; c < CA_D; c += 8) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = o[c + e] * inv;
    *reinterpret_cast<uint4*>(dst + c) = pack8<T>(f);
  }
}

}  // namespace

extern "C" int dbir_clip_embed(const long long* tokens, const float* tok_emb, const float* pos, float* x, int B, int L,
                               int W, int vocab, void* stream) {
  DBIR_CHECK_ARG(tokens && tok_emb && pos && x, "dbir_clip_embed: null pointer");
  DBIR_CHECK_ARG(B > 0 && L > 0 && W > 0 && W % 4 == 0 && vocab > 0, "dbir_clip_embed: bad shape");
  hipLaunchKernelGGL(clip_embed_kernel, dim3((unsigned)(B * L)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), tokens,
                     tok_emb, pos, x, L, W, vocab);
  DBIR_CHECK_LAUNCH("dbir_clip_embed");
  return DBIR_OK;
}

extern "C" int dbir_add_layernorm_f32(int dtype, float* x, const float* y, const float* gamma, const float* beta,
                                      void* out, long long ldo, int out_f32, int rows, int C, float eps, void* stream) {
  DBIR_CHECK_ARG(x && gamma && beta && out, "dbir_add_layernorm_f32: null pointer");
  DBIR_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && ldo >= C && ldo % 4 == 0, "dbir_add_layernorm_f32: bad rows / C / ldo");
  DBIR_CHECK_ARG(dtype == DBIR_F16 || dtype == DBIR_BF16, "dbir_add_layernorm_f32: bad dtype");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == DBIR_F16)
    hipLaunchKernelGGL(add_ln_f32_kernel<F16>, grid, dim3(256), 0, s, x, y, gamma, beta, out, ldo, out_f32, rows, C, eps);
  else
    hipLaunchKernelGGL(add_ln_f32_kernel<BF16>, grid, dim3(256), 0, s, x, y, gamma, beta, out, ldo, out_f32, rows, C, eps);
  DBIR_CHECK_LAUNCH("dbir_add_layernorm_f32");
  return DBIR_OK;
}

extern "C" int dbir_causal_attention(int dtype, const void* qkv, long long ld, void* out, long long ldo, int B, int H,
                                     int L, float scale, void* stream) {
  DBIR_CHECK_ARG(qkv && out, "dbir_causal_attention: null pointer");
  DBIR_CHECK_ARG(B > 0 && H > 0 && L > 0 && L <= CA_MAXL, "dbir_causal_attention: sequence length 1..%d supported (got %d)",
                 CA_MAXL, L);
  DBIR_CHECK_ARG(ld % 8 == 0 && ldo % 8 == 0 && ld >= 3LL * H * CA_D && ldo >= (long long)H * CA_D,
                 "dbir_causal_attention: bad leading dimensions");
  DBIR_CHECK_ARG(((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                 "dbir_causal_attention: qkv / out must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)H, (unsigned)B);
  if (dtype == DBIR_F16)
    hipLaunchKernelGGL(causal_attn_kernel<F16>, grid, dim3(CA_MAXL), 0, s, (const u16*)qkv, ld, (u16*)out, ldo, H, L, scale);
  else if (dtype == DBIR_BF16)
    hipLaunchKernelGGL(causal_attn_kernel<BF16>, grid, dim3(CA_MAXL), 0, s, (const u16*)qkv, ld, (u16*)out, ldo, H, L, scale);
  else {
    dbir_set_error("dbir_causal_attention: bad dtype");
    return DBIR_ERR_ARG;
  }
  DBIR_CHECK_LAUNCH("dbir_causal_attention");
  return DBIR_OK;
}
