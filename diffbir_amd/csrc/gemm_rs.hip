// Register-streaming linear kernel (round 6, tiles 93 - 97): C = A W^T (+ bias, + residual) for the small-M / small-K
// projections of the 16x16 and 8x8 latent levels (reference attention.py:189-216 to_q / to_k / to_v / to_out, 19-45
// FeedForward) where an output tile per CU has too little work to amortise an LDS ring: M = 4096, N = K = 1280 ran at
// 0.15 of the matrix peak through the direct-to-LDS kernels (37 us, 44 launches per evaluation).
//
// What the Gate-A probes of this round established (profiles/r6_ff64_gateA_v1.txt): 16-byte buffer loads straight into
// registers, a few k-steps ahead, sustain 1.5 - 1.9 PF/s of MFMA work per chip as long as the operands sit in L2 / MALL —
// no LDS staging, no ring bookkeeping, no per-tile barrier.  Here BOTH operands stream that way:
//   * eight waves; a wave owns 64 rows x 80 columns as 4 x 5 blocks of v_mfma_f32_16x16x32 (80 accumulator registers);
//     per k-step PAIR (64 k) it loads 8 A pieces and 10 W pieces of 1 KB: a lane reads 32 CONTIGUOUS bytes of one row
//     (k0 + 16 lg .. + 16: the first 16 bytes feed the first MFMA of the pair, the second 16 the second — any assignment
//     of k indices to lanes is valid as long as A and W use the same one), so the four lane groups of a row cover one
//     whole 128-byte line of the ROW-MAJOR operands: no packing, full-line requests;
//   * a workgroup's waves tile (RG x 64 rows) x (CG x 80 columns) x KSP k-slices with RG CG KSP = 8: the k-slices of a
//     tile are summed through LDS in slice order (deterministic), then slice 0 applies the epilogue;
//   * one k-step pair of prefetch (72 registers) — the partner wave on the SIMD covers the rest.
#include <stdint.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int rs_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int rs_u32x2;

struct RsParams {
  const u16* A; long long lda;
  const u16* W; int Kpad;
  const float* bias;
  const u16* R; long long ldr;
  u16* C; long long ldc;
  int M, N, K;
  int tiles_n, ntiles, q;  // column tiles, tiles, tiles per XCD
};

__device__ __forceinline__ int rs_opq(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

template <typename T, int RG, int CG>
__global__ __launch_bounds__(512) void gemm_rs_kernel(const RsParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using vec8 = typename T::vec8;
  constexpr int KSP = 8 / (RG * CG), BM = 64 * RG, BN = 80 * CG;
  static_assert(RG * CG * KSP == 8, "");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave / (RG * CG), rem = wave % (RG * CG), rg = rem / CG, cg = rem % CG;
  const int lr = lane & 15, lg = lane >> 4;
  // consecutive tiles (same A rows, neighbouring W columns) on one XCD
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int t = xcd * p.q + loc;
  if (loc >= p.q || t >= p.ntiles) return;
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int m0 = tm * BM + 64 * rg, n0 = tn * BN + 80 * cg;

  const __amdgpu_buffer_rsrc_t a_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.A), 0, (int)((((long long)p.M - 1) * p.lda + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.W), 0, (int)((long long)p.N * p.Kpad * 2), 0x00020000);
  const int DS = p.K >> 6;                                   // k-step pairs
  const int ds0 = kg * DS / KSP, ds1 = (kg + 1) * DS / KSP;  // this wave's slice
  const int avo = (int)((lr * p.lda + 16 * lg) * 2), wvo = (lr * p.Kpad + 16 * lg) * 2;
  int aso = (int)(((long long)m0 * p.lda + 64 * ds0) * 2), wso = (int)(((long long)n0 * p.Kpad + 64 * ds0) * 2);
  const int a16 = (int)(16 * p.lda * 2), w16 = 16 * p.Kpad * 2;

  f32x4 acc[4][5];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  vec8 fa[2][4][2], fw[2][5][2];  // [register set][block][half of the k-step pair]

  auto load_set = [&](auto set_) __attribute__((always_inline)) {
    constexpr int set = decltype(set_)::value;
    const int av = rs_opq(avo), wv = rs_opq(wvo);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        fa[set][rb][h] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(a_srd, av + 16 * h, aso + rb * a16, 0));
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        fw[set][j][h] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(w_srd, wv + 16 * h, wso + j * w16, 0));
    aso += 128;
    wso += 128;
    asm volatile("" : "+s"(aso), "+s"(wso));
  };
  auto mma_set = [&](auto set_) __attribute__((always_inline)) {
    constexpr int set = decltype(set_)::value;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][j] = T::mfma16(fw[set][j][h], fa[set][rb][h], acc[rb][j]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (ds0 < ds1) load_set(I0{});
  for (int ds = ds0; ds < ds1; ds += 2) {
    if (ds + 1 < ds1) load_set(I1{});
    __builtin_amdgcn_sched_barrier(0);
    mma_set(I0{});
    __builtin_amdgcn_sched_barrier(0);
    if (ds + 1 < ds1) {
      if (ds + 2 < ds1) load_set(I0{});
      __builtin_amdgcn_sched_barrier(0);
      mma_set(I1{});
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- sum the k-slices of a tile in slice order: slices 1 .. KSP - 1 through LDS, slice 0 adds and owns the epilogue
  if constexpr (KSP > 1) {
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (kg > 0) {
      f32x4* dst = red + ((kg - 1) * (RG * CG) + rem) * (20 * 64) + lane;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int j = 0; j < 5; ++j) dst[(rb * 5 + j) * 64] = acc[rb][j];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int s = 0; s < KSP - 1; ++s) {
      const f32x4* src = red + (s * (RG * CG) + rem) * (20 * 64) + lane;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[rb][j] += src[(rb * 5 + j) * 64];
    }
  }
  // ---- epilogue: + bias, + residual, 16-bit rows (a lane holds 4 consecutive columns of one row: 8-byte accesses)
  const __amdgpu_buffer_rsrc_t c_srd = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((((long long)p.M - 1) * p.ldc + p.N) * 2), 0x00020000);
  if (p.bias) {
    const __amdgpu_buffer_rsrc_t b_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.N * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_srd, rs_opq(lg * 16) + 64 * j, n0 * 4, 0));
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][j] += b;
    }
  }
  if (p.R) {
    const __amdgpu_buffer_rsrc_t r_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.R), 0, (int)((((long long)p.M - 1) * p.ldr + p.N) * 2), 0x00020000);
    const int rvo = rs_opq((int)((lr * p.ldr + 4 * lg) * 2));
    const int rso = (int)(((long long)m0 * p.ldr + n0) * 2);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const auto hv = __builtin_amdgcn_raw_buffer_load_b64(r_srd, rvo + 32 * j, rso + (int)(rb * 16 * p.ldr * 2), 0);
        acc[rb][j][0] += T::to_f32((u16)(hv[0] & 0xffff));
        acc[rb][j][1] += T::to_f32((u16)(hv[0] >> 16));
        acc[rb][j][2] += T::to_f32((u16)(hv[1] & 0xffff));
        acc[rb][j][3] += T::to_f32((u16)(hv[1] >> 16));
      }
  }
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    const int vo = rs_opq((int)((((long long)(m0 + 16 * rb + lr) * p.ldc) + n0 + 4 * lg) * 2));
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const rs_u32x2 v = {T::pack2(acc[rb][j][0], acc[rb][j][1]), T::pack2(acc[rb][j][2], acc[rb][j][3])};
      __builtin_amdgcn_raw_buffer_store_b64(v, c_srd, vo + 32 * j, 0, 0);
    }
  }
#endif
}

struct RsShape { int rg, cg; };
bool rs_shape(int tile, RsShape* s) {
  switch (tile) {
    case 93: *s = {2, 2}; return true;  // 128 x 160, 2 k-slices
    case 94: *s = {1, 2}; return true;  //  64 x 160, 4 k-slices
    case 95: *s = {1, 1}; return true;  //  64 x  80, 8 k-slices
    case 96: *s = {2, 4}; return true;  // 128 x 320
    case 97: *s = {4, 2}; return true;  // 256 x 160
  }
  return false;
}

template <typename T, int RG, int CG>
int rs_launch(const RsParams& p, int grid, hipStream_t s) {
  constexpr int KSP = 8 / (RG * CG);
  constexpr int LDS = (KSP - 1) * (RG * CG) * 20 * 64 * 16;
  static bool attr_set = false;
  if (!attr_set && LDS > 0) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rs_kernel<T, RG, CG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      dbir_set_error("dbir_gemm: cannot reserve %d bytes of LDS for the register-streaming kernel", LDS);
      return DBIR_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_rs_kernel<T, RG, CG>), dim3(grid), dim3(512), LDS, s, p);
  return DBIR_OK;
}

}  // namespace

bool dbir_gemm_rs_eligible(const dbir_gemm_desc& d, int tile) {
  RsShape sh;
  if (!rs_shape(tile, &sh)) return false;
  if (d.mode != DBIR_MODE_LINEAR || d.batch > 1 || d.splitk > 1 || d.out_f32 || d.store_mode != 0 || d.rowvec || d.act != DBIR_ACT_NONE ||
      d.out_scale != 1.0f)
    return false;
  if (d.K % 64 || d.M % (64 * sh.rg) || d.N % (80 * sh.cg) || d.N > d.Wrows) return false;
  if (d.lda % 8 || d.ldc % 4 || (d.R && d.ldr % 4)) return false;
  if ((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.W)) & 15) return false;
  if ((reinterpret_cast<uintptr_t>(d.C) | reinterpret_cast<uintptr_t>(d.R)) & 7) return false;
  if (d.bias && (reinterpret_cast<uintptr_t>(d.bias) & 15)) return false;
  const long long lim = 0x7ffffe00LL;
  if (((long long)d.M - 1) * d.lda * 2 + d.K * 2 >= lim || (long long)d.N * d.Kpad * 2 >= lim || ((long long)d.M - 1) * d.ldc * 2 + d.N * 2 >= lim ||
      (d.R && ((long long)d.M - 1) * d.ldr * 2 + d.N * 2 >= lim))
    return false;
  return true;
}

int dbir_gemm_rs(const dbir_gemm_desc& d, int tile, hipStream_t s) {
  RsShape sh;
  rs_shape(tile, &sh);
  RsParams p;
  p.A = (const u16*)d.A; p.lda = d.lda;
  p.W = (const u16*)d.W; p.Kpad = d.Kpad;
  p.bias = d.bias;
  p.R = (const u16*)d.R; p.ldr = d.ldr;
  p.C = (u16*)d.C; p.ldc = d.ldc;
  p.M = d.M; p.N = d.N; p.K = d.K;
  p.tiles_n = d.N / (80 * sh.cg);
  p.ntiles = (d.M / (64 * sh.rg)) * p.tiles_n;
  p.q = cdiv(p.ntiles, 8);
  const int grid = 8 * p.q;
  int rc;
  const bool f16 = d.dtype == DBIR_F16;
#define RS_CASE(RG_, CG_)                                                                      \
  if (sh.rg == RG_ && sh.cg == CG_) {                                                           \
    rc = f16 ? rs_launch<F16, RG_, CG_>(p, grid, s) : rs_launch<BF16, RG_, CG_>(p, grid, s);   \
  } else
  RS_CASE(2, 2) RS_CASE(1, 2) RS_CASE(1, 1) RS_CASE(2, 4) RS_CASE(4, 2) { rc = DBIR_ERR_ARG; }
#undef RS_CASE
  if (rc != DBIR_OK) return rc;
  DBIR_CHECK_LAUNCH("dbir_gemm (register-streaming kernel)");
  return DBIR_OK;
}
