// Layout conversion, elementwise, sampler-step, tiling and pre/post-processing kernels (all HBM-bound).
#include "common.h"

namespace {

constexpr int TPB = 256;
static inline int grid_for(long long n, int cap = 8192) {
  long long g = (n + TPB - 1) / TPB;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

template <typename T>
__global__ void add_scaled_kernel(const u16* __restrict__ a, long long lda, const u16* __restrict__ b,
                                  long long ldb, float s, u16* __restrict__ out, long long ldo, long long M,
                                  int CV) {
  GRID_STRIDE(i, M * CV) {
    const long long m = i / CV;
    const int cv = (int)(i - m * CV);
    float fa[8], fb[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(a + m * lda + cv * 8), fa);
    unpack8<T>(*reinterpret_cast<const uint4*>(b + m * ldb + cv * 8), fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) fa[e] += s * fb[e];
    *reinterpret_cast<uint4*>(out + m * ldo + cv * 8) = pack8<T>(fa);
  }
}

// 2x2 pixel-block <-> channel regrouping of NHWC 16-bit tensors (dtype-agnostic 16-byte moves):
//   to_depth = 1: dst[b, i, j, (ky*2 + kx)*C + c] = src[b, 2i+ky, 2j+kx, c]     ([B,2h,2w,C] -> [B,h,w,4C])
//   to_depth = 0: dst[b, 2i+ky, 2j+kx, c] = src[b, i, j, (ky*2 + kx)*C + c]     ([B,h,w,4C] -> [B,2h,2w,C])
// so that Conv2d(k=2, s=2) and ConvTranspose2d(k=2, s=2) (SCUNet down / up sampling, scunet.py:179-212) become plain
// GEMMs with K = 4C resp. N = 4C.  h, w: the LOW-resolution extent; lds / ldd row strides in elements.
__global__ void block2x2_kernel(const u16* __restrict__ src, long long lds, u16* __restrict__ dst, long long ldd,
                                int B, int h, int w, int CV, int to_depth) {
  GRID_STRIDE(i, (long long)B * h * w * 4 * CV) {
    const int cv = (int)(i % CV);
    long long r = i / CV;
    const int q = (int)(r & 3);
    r >>= 2;
    const int jj = (int)(r % w);
    r /= w;
    const int ii = (int)(r % h);
    const long long b = r / h;
    const long long hi = (b * 2 * h + 2 * ii + (q >> 1)) * (2LL * w) + 2 * jj + (q & 1);  // high-resolution pixel
    const long long lo = (b * h + ii) * (long long)w + jj;                                  // low-resolution pixel
    const long long ch = (long long)q * CV * 8 + cv * 8;
    if (to_depth)
      *reinterpret_cast<uint4*>(dst + lo * ldd + ch) = *reinterpret_cast<const uint4*>(src + hi * lds + cv * 8);
    else
      *reinterpret_cast<uint4*>(dst + hi * ldd + cv * 8) = *reinterpret_cast<const uint4*>(src + lo * lds + ch);
  }
}

// NCHW f32 (one or two sources concatenated along C) -> NHWC 16-bit, zero padded to Cpad (Cpad <= 16 typical)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ s0, int C0, const float* __restrict__ s1, int C1,
                                    u16* __restrict__ dst, int Cpad, int B, long long HW, float scale,
                                    float shift) {
  GRID_STRIDE(i, (long long)B * HW) {
    const long long b = i / HW, p = i - b * HW;
    u16* o = dst + i * Cpad;
    for (int c = 0; c < Cpad; ++c) {
      float v = 0.f;
      if (c < C0)
        v = s0[(b * C0 + c) * HW + p] * scale + shift;
      else if (c < C0 + C1)
        v = s1[(b * C1 + (c - C0)) * HW + p] * scale + shift;
      o[c] = T::from_f32(v);
    }
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ src, int src_f32, long long ld, float* __restrict__ dst,
                                    int C, int B, long long HW, float scale, const float* __restrict__ shift) {
  GRID_STRIDE(i, (long long)B * C * HW) {
    const long long p = i % HW;
    const long long bc = i / HW;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    const long long off = (b * HW + p) * ld + c;
    const float v = src_f32 ? reinterpret_cast<const float*>(src)[off] : T::to_f32(reinterpret_cast<const u16*>(src)[off]);
    dst[i] = v * scale + (shift ? shift[c] : 0.f);
  }
}

// SwinIR front end: out[b, y, x, c*r*r + dy*r + dx] = (in[b, c, y*r+dy, x*r+dx] - mean[c]) * range
template <typename T>
__global__ void pixel_unshuffle_kernel(const float* __restrict__ src, u16* __restrict__ dst, int B, int C, int H,
                                       int W, int r, int Cpad, const float* __restrict__ mean, float range) {
  const int Ho = H / r, Wo = W / r;
  GRID_STRIDE(i, (long long)B * Ho * Wo * Cpad) {
    const int oc = (int)(i % Cpad);
    const long long pix = i / Cpad;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const long long b = pix / ((long long)Wo * Ho);
    float v = 0.f;
    if (oc < C * r * r) {
      const int c = oc / (r * r), rem = oc - c * r * r;
      const int dy = rem / r, dx = rem - dy * r;
      v = (src[((b * C + c) * H + (oy * r + dy)) * (long long)W + ox * r + dx] - mean[c]) * range;
    }
    dst[i] = T::from_f32(v);
  }
}

template <typename T>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, u16* __restrict__ out, int B, int dim,
                                          float max_period) {
  const int half = dim / 2;
  GRID_STRIDE(i, (long long)B * dim) {
    const int b = (int)(i / dim), j = (int)(i - (long long)b * dim);
    float v = 0.f;
    if (j < 2 * half) {
      const int k = j < half ? j : j - half;
      // same arithmetic as the reference: exp(-ln(P) * k / half) in f32, then t * freq, then cos/sin
      const float freq = expf(-logf(max_period) * (float)k / (float)half);
      const float arg = t[b] * freq;
      v = j < half ? cosf(arg) : sinf(arg);
    }
    out[i] = T::from_f32(v);
  }
}

__global__ void lincomb4_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                const float* __restrict__ w, const float* __restrict__ ca,
                                const float* __restrict__ cb, const float* __restrict__ cc,
                                const float* __restrict__ cd, float* __restrict__ out, int B, long long n) {
  GRID_STRIDE(i, (long long)B * n) {
    const int b = (int)(i / n);
    float v = ca[b] * x[i];
    if (y) v += cb[b] * y[i];
    if (z) v += cc[b] * z[i];
    if (w) v += cd[b] * w[i];
    out[i] = v;
  }
}

// Arithmetic order mirrors spaced_sampler.py:158,136-142/128-134,118-126,181-183 (all f32).
__global__ void spaced_step_kernel(const float* __restrict__ x, const float* __restrict__ oc,
                                   const float* __restrict__ ou, const float* __restrict__ noise, float s,
                                   const float* __restrict__ k_x, const float* __restrict__ k_o,
                                   const float* __restrict__ c1, const float* __restrict__ c2,
                                   const float* __restrict__ sd, float* __restrict__ out, int B, long long n) {
  GRID_STRIDE(i, (long long)B * n) {
    const int b = (int)(i / n);
    float o = oc[i];
    if (ou) {
      const float u = ou[i];
      o = u + s * (o - u);
    }
    const float xv = x[i];
    const float x0 = k_x[b] * xv - k_o[b] * o;
    const float mean = c1[b] * x0 + c2[b] * xv;
    out[i] = mean + sd[b] * noise[i];
  }
}

__global__ void tile_gather_kernel(const float* __restrict__ x, float* __restrict__ tiles,
                                   const int* __restrict__ coords, int T, int B, int C, int H, int W, int ts) {
  const long long per = (long long)C * ts * ts;
  GRID_STRIDE(i, (long long)T * B * per) {
    const long long tb = i / per;
    const long long r = i - tb * per;
    const int t = (int)(tb / B), b = (int)(tb - (long long)t * B);
    const int c = (int)(r / (ts * ts));
    const int yy = (int)((r / ts) % ts), xx = (int)(r % ts);
    const int hi = coords[2 * t], wi = coords[2 * t + 1];
    tiles[i] = x[(((long long)b * C + c) * H + hi + yy) * W + wi + xx];
  }
}

__global__ void tile_accumulate_kernel(const float* __restrict__ tiles, const float* __restrict__ weights,
                                       const int* __restrict__ coords, float* __restrict__ out, int T, int B, int C,
                                       int H, int W, int ts) {
  GRID_STRIDE(i, (long long)B * C * H * W) {
    const int xx = (int)(i % W);
    const int yy = (int)((i / W) % H);
    const long long bc = i / ((long long)W * H);
    const int c = (int)(bc % C), b = (int)(bc / C);
    float acc = 0.f, cnt = 0.f;
    for (int t = 0; t < T; ++t) {  // increasing t == the reference's sequential accumulation order
      const int ly = yy - coords[2 * t], lx = xx - coords[2 * t + 1];
      if (ly >= 0 && ly < ts && lx >= 0 && lx < ts) {
        const float w = weights[ly * ts + lx];
        acc += tiles[((((long long)t * B + b) * C + c) * ts + ly) * ts + lx] * w;
        cnt += w;
      }
    }
    out[i] = acc / cnt;
  }
}

// Sharded form of tile_accumulate_kernel: un-normalised weighted sum over a SUBSET of tiles (tiles == nullptr: the
// weights alone, i.e. the normaliser, which depends on the window coordinates only).
__global__ void tile_accumulate_partial_kernel(const float* __restrict__ tiles, const float* __restrict__ weights,
                                               const int* __restrict__ coords, float* __restrict__ num, int T, int B,
                                               int C, int H, int W, int ts) {
  GRID_STRIDE(i, (long long)B * C * H * W) {
    const int xx = (int)(i % W);
    const int yy = (int)((i / W) % H);
    const long long bc = i / ((long long)W * H);
    const int c = (int)(bc % C), b = (int)(bc / C);
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
      const int ly = yy - coords[2 * t], lx = xx - coords[2 * t + 1];
      if (ly >= 0 && ly < ts && lx >= 0 && lx < ts) {
        const float w = weights[ly * ts + lx];
        acc += tiles ? tiles[((((long long)t * B + b) * C + c) * ts + ly) * ts + lx] * w : w;
      }
    }
    num[i] = acc;
  }
}

__global__ void tile_normalize_kernel(const float* __restrict__ num, const float* __restrict__ den,
                                      float* __restrict__ out, long long BC, long long HW) {
  GRID_STRIDE(i, BC * HW) { out[i] = num[i] / den[i % HW]; }
}

__global__ void u8_to_f32_nchw_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int B,
                                      long long HW) {
  GRID_STRIDE(i, (long long)B * 3 * HW) {
    const long long p = i % HW;
    const long long bc = i / HW;
    const int c = (int)(bc % 3);
    const long long b = bc / 3;
    // correctly rounded x/255 (== torch .div(255)): the f64 quotient rounds to the same f32 for all 256 inputs,
    // independent of the compiler's f32 division lowering
    const float v = (float)((double)src[(b * HW + p) * 3 + c] / 255.0);
    dst[i] = fminf(fmaxf(v, 0.f), 1.f);
  }
}

__global__ void wavelet_blur_kernel(const float* __restrict__ src, float* __restrict__ dst, int P, int H, int W,
                                    int radius) {
  GRID_STRIDE(i, (long long)P * H * W) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const float* pl = src + (i / ((long long)W * H)) * (long long)H * W;
    const float kw[3] = {0.25f, 0.5f, 0.25f};
    // conv2d accumulates taps in row-major (ky, kx) order; keep that order
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int sy = min(max(y + (ky - 1) * radius, 0), H - 1);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int sx = min(max(x + (kx - 1) * radius, 0), W - 1);
        acc += pl[(long long)sy * W + sx] * (kw[ky] * kw[kx]);
      }
    }
    dst[i] = acc;
  }
}

__global__ void colorfix_kernel(const float* __restrict__ c, const float* __restrict__ cl,
                                const float* __restrict__ sl, float* __restrict__ out, long long n) {
  GRID_STRIDE(i, n) out[i] = (c[i] - cl[i]) + sl[i];
}

__global__ void f32_nchw_to_u8_nhwc_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, int B,
                                           long long HW) {
  GRID_STRIDE(i, (long long)B * HW * 3) {
    const int c = (int)(i % 3);
    const long long bp = i / 3;
    const long long b = bp / HW, p = bp - b * HW;
    float v = src[(b * 3 + c) * HW + p] * 255.0f;
    v = fminf(fmaxf(v, 0.f), 255.f);
    dst[i] = (unsigned char)v;  // truncation, like tensor.to(torch.uint8)
  }
}

}  // namespace

#define STREAM reinterpret_cast<hipStream_t>(stream)
#define BY_DTYPE(KERNEL, grid, ...)                                                     \
  do {                                                                                  \
    if (dtype == DBIR_F16)                                                              \
      hipLaunchKernelGGL((KERNEL<F16>), dim3(grid), dim3(TPB), 0, STREAM, __VA_ARGS__); \
    else if (dtype == DBIR_BF16)                                                        \
      hipLaunchKernelGGL((KERNEL<BF16>), dim3(grid), dim3(TPB), 0, STREAM, __VA_ARGS__); \
    else {                                                                              \
      dbir_set_error(#KERNEL ": bad dtype %d", dtype);                                  \
      return DBIR_ERR_ARG;                                                              \
    }                                                                                   \
  } while (0)

extern "C" int dbir_add_scaled(int dtype, const void* a, long long lda, const void* b, long long ldb, float s,
                               void* out, long long ldo, long long M, int C, void* stream) {
  DBIR_CHECK_ARG(a && b && out && C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0 && M > 0,
                 "dbir_add_scaled: bad args");
  BY_DTYPE(add_scaled_kernel, grid_for(M * (C / 8)), (const u16*)a, lda, (const u16*)b, ldb, s, (u16*)out, ldo, M,
           C / 8);
  DBIR_CHECK_LAUNCH("dbir_add_scaled");
  return DBIR_OK;
}

extern "C" int dbir_block2x2(const void* src, long long lds, void* dst, long long ldd, int B, int h, int w, int C,
                             int to_depth, void* stream) {
  DBIR_CHECK_ARG(src && dst && B > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0,
                 "dbir_block2x2: bad args (C and the row strides must be multiples of 8)");
  DBIR_CHECK_ARG(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0,
                 "dbir_block2x2: 16-byte aligned tensors");
  DBIR_CHECK_ARG(to_depth ? (lds >= C && ldd >= 4LL * C) : (lds >= 4LL * C && ldd >= C), "dbir_block2x2: bad row strides");
  hipLaunchKernelGGL(block2x2_kernel, grid_for((long long)B * h * w * 4 * (C / 8)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const u16*)src, lds, (u16*)dst, ldd, B, h, w, C / 8, to_depth);
  DBIR_CHECK_LAUNCH("dbir_block2x2");
  return DBIR_OK;
}

extern "C" int dbir_nchw_to_nhwc(int dtype, const float* src0, int C0, const float* src1, int C1, void* dst,
                                 int Cpad, int B, int H, int W, float scale, float shift, void* stream) {
  DBIR_CHECK_ARG(src0 && dst && C0 > 0 && C0 + C1 <= Cpad && (C1 == 0 || src1), "dbir_nchw_to_nhwc: bad args");
  BY_DTYPE(nchw_to_nhwc_kernel, grid_for((long long)B * H * W), src0, C0, src1, C1, (u16*)dst, Cpad, B,
           (long long)H * W, scale, shift);
  DBIR_CHECK_LAUNCH("dbir_nchw_to_nhwc");
  return DBIR_OK;
}

// 16-byte row copies between strided 16-bit matrices
__global__ void copy_rows_kernel(const uint4* __restrict__ src, long long lds8, uint4* __restrict__ dst, long long ldd8,
                                 long long M, int C8) {
  GRID_STRIDE(i, M * C8) {
    const long long m = i / C8;
    const int c = (int)(i - m * C8);
    dst[m * ldd8 + c] = src[m * lds8 + c];
  }
}

extern "C" int dbir_copy_rows(const void* src, long long lds, void* dst, long long ldd, long long M, int C, void* stream) {
  DBIR_CHECK_ARG(src && dst && M > 0 && C > 0 && C % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && lds >= C && ldd >= C &&
                     (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
                 "dbir_copy_rows: need C, lds, ldd multiples of 8 and 16-byte aligned pointers");
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(M * (C / 8))), dim3(TPB), 0, STREAM, reinterpret_cast<const uint4*>(src),
                     lds / 8, reinterpret_cast<uint4*>(dst), ldd / 8, M, C / 8);
  DBIR_CHECK_LAUNCH("dbir_copy_rows");
  return DBIR_OK;
}

extern "C" int dbir_nhwc_to_nchw(int dtype, const void* src, int src_f32, long long ld, float* dst, int C, int B,
                                 int H, int W, float scale, const float* shift, void* stream) {
  DBIR_CHECK_ARG(src && dst && C > 0 && ld >= C, "dbir_nhwc_to_nchw: bad args");
  BY_DTYPE(nhwc_to_nchw_kernel, grid_for((long long)B * C * H * W), src, src_f32, ld, dst, C, B, (long long)H * W,
           scale, shift);
  DBIR_CHECK_LAUNCH("dbir_nhwc_to_nchw");
  return DBIR_OK;
}

extern "C" int dbir_pixel_unshuffle(int dtype, const float* src, void* dst, int B, int C, int H, int W, int r,
                                    int Cpad, const float* mean, float range, void* stream) {
  DBIR_CHECK_ARG(src && dst && mean && H % r == 0 && W % r == 0 && Cpad >= C * r * r, "dbir_pixel_unshuffle: bad args");
  BY_DTYPE(pixel_unshuffle_kernel, grid_for((long long)B * (H / r) * (W / r) * Cpad), src, (u16*)dst, B, C, H, W, r,
           Cpad, mean, range);
  DBIR_CHECK_LAUNCH("dbir_pixel_unshuffle");
  return DBIR_OK;
}

extern "C" int dbir_timestep_embedding(int dtype, const float* t, void* out, int B, int dim, float max_period,
                                       void* stream) {
  DBIR_CHECK_ARG(t && out && B > 0 && dim > 1, "dbir_timestep_embedding: bad args");
  BY_DTYPE(timestep_embedding_kernel, grid_for((long long)B * dim), t, (u16*)out, B, dim, max_period);
  DBIR_CHECK_LAUNCH("dbir_timestep_embedding");
  return DBIR_OK;
}

extern "C" int dbir_lincomb4(const float* x, const float* y, const float* z, const float* w, const float* ca,
                             const float* cb, const float* cc, const float* cd, float* out, int B, long long n,
                             void* stream) {
  DBIR_CHECK_ARG(x && ca && out && (!y || cb) && (!z || cc) && (!w || cd), "dbir_lincomb4: bad args");
  hipLaunchKernelGGL(lincomb4_kernel, dim3(grid_for((long long)B * n)), dim3(TPB), 0, STREAM, x, y, z, w, ca, cb, cc,
                     cd, out, B, n);
  DBIR_CHECK_LAUNCH("dbir_lincomb4");
  return DBIR_OK;
}

extern "C" int dbir_spaced_step(const float* x, const float* oc, const float* ou, const float* noise, float s,
                                const float* k_x, const float* k_o, const float* c1, const float* c2,
                                const float* sd, float* out, int B, long long n, void* stream) {
  DBIR_CHECK_ARG(x && oc && noise && k_x && k_o && c1 && c2 && sd && out, "dbir_spaced_step: null pointer");
  hipLaunchKernelGGL(spaced_step_kernel, dim3(grid_for((long long)B * n)), dim3(TPB), 0, STREAM, x, oc, ou, noise, s,
                     k_x, k_o, c1, c2, sd, out, B, n);
  DBIR_CHECK_LAUNCH("dbir_spaced_step");
  return DBIR_OK;
}

extern "C" int dbir_tile_gather(const float* x, float* tiles, const int* coords, int T, int B, int C, int H, int W,
                                int ts, void* stream) {
  DBIR_CHECK_ARG(x && tiles && coords && T > 0 && ts <= H && ts <= W, "dbir_tile_gather: bad args");
  hipLaunchKernelGGL(tile_gather_kernel, dim3(grid_for((long long)T * B * C * ts * ts)), dim3(TPB), 0, STREAM, x,
                     tiles, coords, T, B, C, H, W, ts);
  DBIR_CHECK_LAUNCH("dbir_tile_gather");
  return DBIR_OK;
}

extern "C" int dbir_tile_accumulate(const float* tiles, const float* weights, const int* coords, float* out, int T,
                                    int B, int C, int H, int W, int ts, void* stream) {
  DBIR_CHECK_ARG(tiles && weights && coords && out && T > 0, "dbir_tile_accumulate: bad args");
  hipLaunchKernelGGL(tile_accumulate_kernel, dim3(grid_for((long long)B * C * H * W)), dim3(TPB), 0, STREAM, tiles,
                     weights, coords, out, T, B, C, H, W, ts);
  DBIR_CHECK_LAUNCH("dbir_tile_accumulate");
  return DBIR_OK;
}

extern "C" int dbir_tile_accumulate_partial(const float* tiles, const float* weights, const int* coords, float* num,
                                            int T, int B, int C, int H, int W, int ts, void* stream) {
  DBIR_CHECK_ARG(weights && coords && num && T > 0, "dbir_tile_accumulate_partial: bad args");
  hipLaunchKernelGGL(tile_accumulate_partial_kernel, dim3(grid_for((long long)B * C * H * W)), dim3(TPB), 0, STREAM,
                     tiles, weights, coords, num, T, B, C, H, W, ts);
  DBIR_CHECK_LAUNCH("dbir_tile_accumulate_partial");
  return DBIR_OK;
}

extern "C" int dbir_tile_normalize(const float* num, const float* den, float* out, long long BC, long long HW,
                                   void* stream) {
  DBIR_CHECK_ARG(num && den && out && BC > 0 && HW > 0, "dbir_tile_normalize: bad args");
  hipLaunchKernelGGL(tile_normalize_kernel, dim3(grid_for(BC * HW)), dim3(TPB), 0, STREAM, num, den, out, BC, HW);
  DBIR_CHECK_LAUNCH("dbir_tile_normalize");
  return DBIR_OK;
}

extern "C" int dbir_u8_to_f32_nchw(const unsigned char* src, float* dst, int B, int H, int W, void* stream) {
  DBIR_CHECK_ARG(src && dst, "dbir_u8_to_f32_nchw: null pointer");
  hipLaunchKernelGGL(u8_to_f32_nchw_kernel, dim3(grid_for((long long)B * 3 * H * W)), dim3(TPB), 0, STREAM, src, dst,
                     B, (long long)H * W);
  DBIR_CHECK_LAUNCH("dbir_u8_to_f32_nchw");
  return DBIR_OK;
}

extern "C" int dbir_wavelet_blur(const float* src, float* dst, int P, int H, int W, int radius, void* stream) {
  DBIR_CHECK_ARG(src && dst && src != dst && radius > 0, "dbir_wavelet_blur: bad args");
  hipLaunchKernelGGL(wavelet_blur_kernel, dim3(grid_for((long long)P * H * W)), dim3(TPB), 0, STREAM, src, dst, P, H,
                     W, radius);
  DBIR_CHECK_LAUNCH("dbir_wavelet_blur");
  return DBIR_OK;
}

extern "C" int dbir_colorfix(const float* content, const float* content_low, const float* style_low, float* out,
                             long long n, void* stream) {
  DBIR_CHECK_ARG(content && content_low && style_low && out, "dbir_colorfix: null pointer");
  hipLaunchKernelGGL(colorfix_kernel, dim3(grid_for(n)), dim3(TPB), 0, STREAM, content, content_low, style_low, out, n);
  DBIR_CHECK_LAUNCH("dbir_colorfix");
  return DBIR_OK;
}

extern "C" int dbir_f32_nchw_to_u8_nhwc(const float* src, unsigned char* dst, int B, int H, int W, void* stream) {
  DBIR_CHECK_ARG(src && dst, "dbir_f32_nchw_to_u8_nhwc: null pointer");
  hipLaunchKernelGGL(f32_nchw_to_u8_nhwc_kernel, dim3(grid_for((long long)B * H * W * 3)), dim3(TPB), 0, STREAM, src,
                     dst, B, (long long)H * W);
  DBIR_CHECK_LAUNCH("dbir_f32_nchw_to_u8_nhwc");
  return DBIR_OK;
}
