// Fused transformer-block kernels, second generation (round 6) — the 64x64 latent level (C = 320) and the 32x32 level
// (C = 640) of the SD-2.1 UNet / ControlNet (reference diffbir/model/attention.py:19-45 GEGLU / FeedForward, 189-216
// CrossAttention, 265-274 BasicTransformerBlock._forward, 334-353 SpatialTransformer.forward).  Same operator boundary
// as csrc/xformer.hip (dbir_xf_head / dbir_xf_tail, include/dbir.h), different machine mapping, chosen from the Gate-A
// measurements of tools/probes/xf_ff64.hip / xf_ff16.hip (profiles/r6_ff64_gateA_v1.txt):
//
//   * a wave owns 64 rows x 80 output columns as 4 x 5 blocks of v_mfma_f32_16x16x32: one weight piece (16 columns x
//     32 k = 1 KB) feeds 4 MFMAs — half the weight bytes per FLOP of the 32-row waves of xformer.hip with 80 accumulator
//     registers, so EIGHT waves (two per SIMD) fit; a lane holds 4 consecutive output columns of one row;
//   * the weights never touch LDS: every wave streams the pieces of ITS column group straight into a 10-piece register
//     ring with 16-byte buffer loads (stream order = consumption order, one running scalar offset; the C = 640 panel has
//     8 column groups = private streams, the C = 320 panel 4 column groups x 2 row groups).  No LDS ring, no
//     direct-to-LDS bookkeeping, no per-tile workgroup barrier; the compiler counts vmcnt;
//   * LDS holds only the activation panel image X [16-row block][k-step of 32][lane][16 B] (80 KB: 128 x 320 or 64 x 640),
//     the double-buffered GEGLU chunk (2 x 16 KB) and the feed-forward bias table;
//   * GEGLU feed-forward: 20 chunks of 16 hidden units per column group; the two wave groups (waves 0-3 / 4-7 = one
//     wave of each on every SIMD) run ONE BARRIER APART, so that while one wave of a SIMD multiplies the projection of a
//     chunk (MFMA-dense) its partner does the GELU arithmetic of the previous one (VALU-dense: VALU and MFMA work of ONE
//     wave do not overlap on this chip, of two waves they do) and its output projection: one workgroup barrier per
//     120 / 240 MFMAs per wave;
//   * gelu(x) = x * sigmoid(x * p(min(x^2, 64))), p fitted to the erf form (max abs error 8.1e-5 = 0.17 f16 ulp at 1):
//     9 VALU instructions instead of 15.
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace {

template <int CC>
struct X2Cfg {
  static_assert(CC == 320 || CC == 640, "fused transformer kernels: C = 320 or 640");
  static constexpr int C = CC, CG = CC / 80, RG = 8 / CG, BM = 64 * RG, KS = CC / 32, HEADS = CC / 64;
  static constexpr int CHH = 16 * CG;                 // hidden units per feed-forward chunk: 16 per column group
  static constexpr int NCH = 4 * CC / CHH;            // chunks (20)
  static constexpr int GK = CHH / 32;                 // k-steps of a chunk's output projection (2 / 4)
  static constexpr int P = 10;                        // weight ring: pieces in flight per wave
  static constexpr int GP = 5 * KS;                   // pieces of a C x C GEMM per column group (50 / 100)
  static constexpr int F1P = 2 * KS, F2P = 5 * GK;    // pieces of a chunk's projection / output projection
  static constexpr int TAIL_PIECES = 4 * GP + NCH * (F1P + F2P);  // out1, q2, out2, feed-forward, proj_out (800 / 1600)
  static constexpr int HEAD_PIECES = 4 * GP;                      // proj_in, q, k, v
  static constexpr int X_BYTES = BM * CC * 2;         // 81920
  static constexpr int GB_BYTES = BM * CHH * 2;       // 16384
  static constexpr int B1_FLOATS = NCH * CG * 32;     // projection bias of every chunk: [chunk][column group][value | gate][16]
  static constexpr int TAIL_LDS = X_BYTES + 2 * GB_BYTES + (B1_FLOATS + 5 * CC) * 4;   // + the 5 bias rows
  static constexpr int HEAD_LDS = X_BYTES + 2 * CG * BM * 4;   // + LayerNorm exchange
  static_assert(GP % P == 0 && F1P % P == 0 && F2P % P == 0, "every sub-block must start at ring slot 0");
};
constexpr int X2NT = 512;
constexpr int X2KB = 6, X2LKP = 96;  // padded text context: 6 key blocks of 16

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct Xf2Params {
  const u16* o;  long long ldo;    // tail: self-attention output [Msrc, C]        head: block input x [M, C]
  const u16* h;  long long ldh;    // tail: residual stream before out1 [Msrc, C]  head: (out) h = proj_in(GN(x)) [M, C]
  const u16* x;  long long ldx;    // tail: block input (residual of proj_out) [Msrc, C]
  u16* out;      long long ldout;  // tail: block output [M, C]                     head: (out) q|k [M, 2C]
  u16* vt;       long long vt_ld, vt_bs;  // head: (out) v^T [B, C, Lpad]
  const float* ab;                 // head: GroupNorm scale | shift per (sample, channel) f32 [B, 2, C]
  int M, L;
  int pair_bs;
  const void* wstream;             // [CG][PIECES + P] pieces of 1 KB
  const float* prm;                // tail: f32 [5][C] bias rows, then the feed-forward projection bias table [NCH][CG][2][16]
  const u16* kf; const u16* vf;    // tail: context K / V^T fragments [B][heads][6][2][64][8], [B][heads][4][3][64][8]
  int Lk;  float c;
  int npanels, q, gx;
  int stop_after;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t x2srd(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7ffffe00LL ? 0x7ffffe00LL : bytes), 0x00020000);
}
__device__ __forceinline__ void x2barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// gelu(x) = x * Phi(x) ~ x / (1 + 2^(x * p(min(x^2, 64)))), p = -log2(e) * (1.5961 + 0.07331 x^2 - 0.000582 x^4)
__device__ __forceinline__ float gelu_sp(float x) {
  const float x2 = fminf(x * x, 64.0f);
  float p = __builtin_fmaf(8.39458781e-04f, x2, -1.05767970e-01f);
  p = __builtin_fmaf(p, x2, -2.30265908f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * x));
}

// A lane-offset VGPR made opaque at its use site: "base + literal" offsets are loop-invariant, LICM hoists every one of them out
// of the panel loop into its own VGPR, they spill, and every scratch reload is a vmcnt(0) that drains the weight ring; behind
// this no-op the literal stays next to the access and folds into its offset field.
__device__ __forceinline__ int x2opq(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
#define X2_IC(N) std::integral_constant<int, (N)>{}
template <int LO, int HI, typename F>
__device__ __forceinline__ void x2_for(F&& fn) {
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) { (fn(std::integral_constant<int, LO + I>{}), ...); }
  (std::make_integer_sequence<int, HI - LO>{});
}

// ===============================================================================================================
// xf2_tail
// ===============================================================================================================
template <typename T, int DBG, int CC>  // DBG 1 = intermediate dumps (tests), 2 = section timing (DBIR_DIAG builds, tools/probes/xf2_anatomy.py)
__global__ __launch_bounds__(X2NT) void xf2_tail_kernel(const Xf2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using G = X2Cfg<CC>;
  using vec8 = typename T::vec8;
  constexpr int C = CC, CG = G::CG, KS = G::KS, GK = G::GK, NCH = G::NCH, BM = G::BM, P = G::P, HEADS = G::HEADS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;  // wave group: one wave of each group on every SIMD
  const int rg = CG == 4 ? wave >> 2 : 0, cg = CG == 4 ? wave & 3 : wave;
  const int lr = lane & 15, lg = lane >> 4;
  const int lane16 = lane * 16;
  // byte offset inside an operand image of the 4 consecutive columns a lane holds of 16-column block t of row block rbg
  // (KST k-steps of 32 per row block): rbg * KST * 1024 + t * 512 + lanew
  const int lanew = (lg >> 1) * 256 + lr * 16 + (lg & 1) * 8;

  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int x0 = xcd * p.q;
  int xn = p.npanels - x0;
  xn = xn > p.q ? p.q : xn;
  const int nmine = xn > loc ? (xn - loc + p.gx - 1) / p.gx : 0;
  if (nmine == 0) return;

  constexpr int SPW = G::TAIL_PIECES + P;
  const __amdgpu_buffer_rsrc_t w_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wstream), 0, CG * SPW * 1024, 0x00020000);
  const long long src_rows = p.pair_bs ? (long long)p.M / 2 : (long long)p.M;
  const __amdgpu_buffer_rsrc_t o_srd = x2srd(p.o, (((src_rows - 1) * p.ldo + C) * 2 + 15) & ~15LL);
  const __amdgpu_buffer_rsrc_t h_srd = x2srd(p.h, ((src_rows - 1) * p.ldh + C) * 2);
  const __amdgpu_buffer_rsrc_t x_srd = x2srd(p.x, ((src_rows - 1) * p.ldx + C) * 2);
  const __amdgpu_buffer_rsrc_t out_srd = x2srd(p.out, (((long long)p.M - 1) * p.ldout + C) * 2);
  const __amdgpu_buffer_rsrc_t prm_srd = x2srd(p.prm, (5 * C + G::B1_FLOATS) * 4);
  const long long nsamp = (long long)p.M / p.L;
  const __amdgpu_buffer_rsrc_t kf_srd = x2srd(p.kf, nsamp * HEADS * 12 * 1024);
  const __amdgpu_buffer_rsrc_t vf_srd = x2srd(p.vf, nsamp * HEADS * 12 * 1024);

  // a column group's stream is consumed strictly in order: ONE running scalar offset, P pieces ahead of the consumer
  const int wsb = cg * (SPW * 1024);
  int wp = wsb;
  auto wload = [&](int j) __attribute__((always_inline)) -> vec8 {
    return __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(w_srd, x2opq(lane16) + (j & 3) * 1024, wp + (j >> 2) * 4096, 0));
  };
#define X2_WP_ADV(N) do { wp += (N) * 1024; asm volatile("" : "+s"(wp)); } while (0)

  vec8 wq[P];
  vec8 xa[4], xb[4];
  f32x4 acc[4][5];   // [16-row block][16-column block]: the wave's 64 x 80 output tile
  f32x4 gacc[4][2];  // [16-row block][value | gate]: the GEGLU projection of the chunk in flight
  uint2 hres[4][5];  // the residual stream, packed 16 bit, same lane layout as acc

  // ring := the first P pieces of the stream
  auto prime = [&]() __attribute__((always_inline)) {
    wp = wsb;
    asm volatile("" : "+s"(wp));
#pragma unroll
    for (int i = 0; i < P; i += 2) {
      wq[i] = wload(0);
      wq[i + 1] = wload(1);
      X2_WP_ADV(2);
    }
  };
  prime();
  char* const gb0 = smem + G::X_BYTES;
  float* const b1l = reinterpret_cast<float*>(smem + G::X_BYTES + 2 * G::GB_BYTES);
  // feed-forward projection bias table and the 5 bias rows -> LDS, once per workgroup (global parameter loads in front of a GEMM
  // were 1 - 2 us of exposed L2 latency each)
  float* const brl = b1l + G::B1_FLOATS;
  for (int i = tid; i < G::B1_FLOATS / 4; i += X2NT)
    reinterpret_cast<f32x4*>(b1l)[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prm_srd, i * 16, 5 * C * 4, 0));
  for (int i = tid; i < 5 * C / 4; i += X2NT)
    reinterpret_cast<f32x4*>(brl)[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prm_srd, i * 16, 0, 0));

  // ---- acc[rb][j] += A[rows of this wave][k] W[columns of this wave][k] over KST k-steps of 32: A fragments from the operand
  //      image at `abase` (this wave's first row block), 5 weight pieces per k-step (ring period = 2 k-steps)
  auto gemm5 = [&](const char* abase, auto kst_) __attribute__((always_inline)) {
    constexpr int KST = decltype(kst_)::value;
    static_assert(KST % 2 == 0, "");
    const char* ap = abase + x2opq(lane16);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) xa[rb] = *reinterpret_cast<const vec8*>(ap + rb * KST * 1024);
    for (int k2 = 0; k2 < KST; k2 += 2) {
      x2_for<0, 2>([&](auto half_) __attribute__((always_inline)) {
        constexpr int half = decltype(half_)::value;
        // fragments of the next k-step into the other register set
        if (half == 0 || k2 + 2 < KST) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) {
            if constexpr (half == 0) xb[rb] = *reinterpret_cast<const vec8*>(ap + rb * KST * 1024 + 1024);
            else xa[rb] = *reinterpret_cast<const vec8*>(ap + rb * KST * 1024 + 2048);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        x2_for<0, 5>([&](auto j_) __attribute__((always_inline)) {
          constexpr int j = decltype(j_)::value, slot = 5 * half + j;
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) acc[rb][j] = T::mfma16(wq[slot], half ? xb[rb] : xa[rb], acc[rb][j]);
          __builtin_amdgcn_sched_barrier(0);
          wq[slot] = wload(j);
          __builtin_amdgcn_sched_barrier(0);
        });
        X2_WP_ADV(5);
      });
      ap += 2048;
    }
  };
  // ---- GEGLU projection of one chunk: K = C, 4 row blocks x (value, gate); ring period = 5 k-steps
  auto f1 = [&](int c) __attribute__((always_inline)) {
    const char* ap = smem + (4 * rg) * KS * 1024 + x2opq(lane16);
    // the chunk's bias (hidden units 4 lg .. + 3 of this wave's 16, value | gate) = the C operand of the first MFMAs
    const f32x4* bl = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(b1l) + (c * CG + cg) * 128 + x2opq(lg * 16));
    const f32x4 bq[2] = {bl[0], bl[4]};
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) xa[rb] = *reinterpret_cast<const vec8*>(ap + rb * KS * 1024);
    x2_for<0, KS>([&](auto ks_) __attribute__((always_inline)) {
      constexpr int ks = decltype(ks_)::value;
      if constexpr (ks + 1 < KS) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          if constexpr (ks & 1) xa[rb] = *reinterpret_cast<const vec8*>(ap + (rb * KS + ks + 1) * 1024);
          else xb[rb] = *reinterpret_cast<const vec8*>(ap + (rb * KS + ks + 1) * 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      x2_for<0, 2>([&](auto nb_) __attribute__((always_inline)) {
        constexpr int nb = decltype(nb_)::value, slot = (2 * ks + nb) % P;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const vec8 xf = (ks & 1) ? xb[rb] : xa[rb];
          if constexpr (ks == 0) {
            gacc[rb][nb] = T::mfma16(wq[slot], xf, bq[nb]);
          } else {
            gacc[rb][nb] = T::mfma16(wq[slot], xf, gacc[rb][nb]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        wq[slot] = wload(nb);
        __builtin_amdgcn_sched_barrier(0);
      });
      X2_WP_ADV(2);
    });
  };
  // ---- g = value * gelu(gate) (biases included by the projection) of the chunk this wave has just finished -> chunk image gbw
  auto gelu = [&](char* gbw) __attribute__((always_inline)) {
    char* const gbo = gbw + (4 * rg) * GK * 1024 + cg * 512 + x2opq(lanew);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gacc[rb][0][e] * gelu_sp(gacc[rb][1][e]);
      uint2 pk;
      pk.x = T::pack2(v[0], v[1]);
      pk.y = T::pack2(v[2], v[3]);
      *reinterpret_cast<uint2*>(gbo + rb * GK * 1024) = pk;
    }
  };
  // ---- output projection of one chunk: K = chunk, 4 row blocks x 5 column blocks
  auto f2 = [&](const char* gbr) __attribute__((always_inline)) {
    const char* ap = gbr + (4 * rg) * GK * 1024 + x2opq(lane16);
    x2_for<0, GK>([&](auto k_) __attribute__((always_inline)) {
      constexpr int k = decltype(k_)::value;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) xa[rb] = *reinterpret_cast<const vec8*>(ap + (rb * GK + k) * 1024);
      __builtin_amdgcn_sched_barrier(0);
      x2_for<0, 5>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value, slot = (5 * k + j) % P;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][j] = T::mfma16(wq[slot], xa[rb], acc[rb][j]);
        __builtin_amdgcn_sched_barrier(0);
        wq[slot] = wload(j);
        __builtin_amdgcn_sched_barrier(0);
      });
      X2_WP_ADV(5);
    });
  };
  // ---- accumulators := f32 parameter row `prow` (+ 16-bit residual rows from `srd`, row stride ld, first row r0)
  auto acc_bias = [&](int prow) __attribute__((always_inline)) {
    const char* bp = reinterpret_cast<const char*>(brl) + (prow * C + 80 * cg) * 4 + x2opq(lg * 16);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 64 * j);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][j] = b;
    }
  };
  // 16-bit residual rows (row stride ld, first row r0) -> hres: issued in FRONT of a GEMM, added behind it (acc_add_hres)
  auto load_rows_hres = [&](__amdgpu_buffer_rsrc_t srd, long long ld, long long r0) __attribute__((always_inline)) {
    const int vo = x2opq((int)((lr * ld + 4 * lg) * 2));
    const int so = (int)(((r0 + 64 * rg) * ld + 80 * cg) * 2);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const auto hv = __builtin_amdgcn_raw_buffer_load_b64(srd, vo + 32 * j, so + (int)(rb * 16 * ld * 2), 0);
        hres[rb][j].x = hv[0];
        hres[rb][j].y = hv[1];
      }
  };
  auto acc_add_hres = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        acc[rb][j][0] += T::to_f32((u16)(hres[rb][j].x & 0xffff));
        acc[rb][j][1] += T::to_f32((u16)(hres[rb][j].x >> 16));
        acc[rb][j][2] += T::to_f32((u16)(hres[rb][j].y & 0xffff));
        acc[rb][j][3] += T::to_f32((u16)(hres[rb][j].y >> 16));
      }
  };
  // residual stream: round the accumulators to 16 bit -> hres (packed) and back into acc as the rounded values
  auto round_to_hres = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        hres[rb][j].x = T::pack2(acc[rb][j][0], acc[rb][j][1]);
        hres[rb][j].y = T::pack2(acc[rb][j][2], acc[rb][j][3]);
        acc[rb][j][0] = T::to_f32((u16)(hres[rb][j].x & 0xffff));
        acc[rb][j][1] = T::to_f32((u16)(hres[rb][j].x >> 16));
        acc[rb][j][2] = T::to_f32((u16)(hres[rb][j].y & 0xffff));
        acc[rb][j][3] = T::to_f32((u16)(hres[rb][j].y >> 16));
      }
  };
  // acc (scaled) -> the operand image X, 16 bit: 20 ds_write_b64 per lane
  auto store_x = [&](const float (&sc)[4]) __attribute__((always_inline)) {
    char* const xo = smem + (4 * rg) * KS * 1024 + (5 * cg) * 512 + x2opq(lanew);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        uint2 pk;
        pk.x = T::pack2(acc[rb][j][0] * sc[rb], acc[rb][j][1] * sc[rb]);
        pk.y = T::pack2(acc[rb][j][2] * sc[rb], acc[rb][j][3] * sc[rb]);
        *reinterpret_cast<uint2*>(xo + rb * KS * 1024 + j * 512) = pk;
      }
  };
  // LayerNorm over the C columns of every panel row: input = acc (f32 values), normalised rows (16 bit) -> X.  No affine map here
  // (folded into the consuming GEMM on the host).  A row is spread over the 4 lane groups of CG waves: two-pass statistics
  // (mean, then centred sum of squares) with one LDS exchange each; the barriers also order the preceding GEMM's last reads
  // of X before the writes below.  Scratch: the (idle) GEGLU chunk buffers.
  auto layernorm_to_x = [&]() __attribute__((always_inline)) {
    float* red = reinterpret_cast<float*>(gb0 + x2opq(lr * 4)) + 64 * rg;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 5; ++j) t += (acc[rb][j][0] + acc[rb][j][1]) + (acc[rb][j][2] + acc[rb][j][3]);
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (lg == 0) red[cg * BM + 16 * rb] = t;
    }
    x2barrier();
    float mean[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float m = 0.f;
#pragma unroll
      for (int w = 0; w < CG; ++w) m += red[w * BM + 16 * rb];
      mean[rb] = m * (1.0f / C);
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[rb][j][e] -= mean[rb];
          t += acc[rb][j][e] * acc[rb][j][e];
        }
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (lg == 0) red[(CG + cg) * BM + 16 * rb] = t;
    }
    x2barrier();
    float rstd[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < CG; ++w) v += red[(CG + w) * BM + 16 * rb];
      rstd[rb] = rsqrtf(v * (1.0f / C) + 1e-5f);
    }
    store_x(rstd);
  };
  // rows out of the accumulators: 4 consecutive columns per lane -> 8-byte stores
  auto store_rows = [&](long long r0) __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int vo = x2opq((int)((((r0 + 64 * rg + 16 * rb + lr) * p.ldout) + 80 * cg + 4 * lg) * 2));
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const u32x2 v = {T::pack2(acc[rb][j][0], acc[rb][j][1]), T::pack2(acc[rb][j][2], acc[rb][j][3])};
        __builtin_amdgcn_raw_buffer_store_b64(v, out_srd, vo + 32 * j, 0, 0);
      }
    }
  };
  // DEBUG dumps (tests): the operand image X, or the packed residual stream, as rows of `out`
  auto dump_x = [&](long long r0) __attribute__((always_inline)) {
    x2barrier();
    for (int q = tid; q < BM * C / 8; q += X2NT) {
      const int piece = q >> 6, l = q & 63;
      const int rbg = piece / KS, ks = piece - rbg * KS;
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + q * 16);
      __builtin_amdgcn_raw_buffer_store_b128(v, out_srd, (int)(((r0 + rbg * 16 + (l & 15)) * p.ldout + ks * 32 + (l >> 4) * 8) * 2), 0, 0);
    }
    x2barrier();
  };
  auto dump_hres = [&](long long r0) __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int vo = x2opq((int)((((r0 + 64 * rg + 16 * rb + lr) * p.ldout) + 80 * cg + 4 * lg) * 2));
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const u32x2 v = {hres[rb][j].x, hres[rb][j].y};
        __builtin_amdgcn_raw_buffer_store_b64(v, out_srd, vo + 32 * j, 0, 0);
      }
    }
  };
  const float one4[4] = {1.f, 1.f, 1.f, 1.f};
  // DEBUG anatomy (DBG == 2): s_memtime per section, summed over this workgroup's panels, written by waves 0 and 4 into `h`
  unsigned long long ta[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  if (DBG == 2) tprev = __builtin_amdgcn_s_memtime();
  auto ts = [&](int i) __attribute__((always_inline)) {
    if constexpr (DBG == 2) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      ta[i] += now - tprev;
      tprev = now;
    }
  };

  for (int pi = 0; pi < nmine; ++pi) {
    const int panel = x0 + loc + pi * p.gx;
    const long long row0 = (long long)panel * BM;
    const int b = (int)(row0 / p.L);  // sample of the (full) batch; a panel never straddles samples (L % BM == 0)
    long long srow0 = row0;
    if (p.pair_bs) {
      const int g2 = b / (2 * p.pair_bs), r2 = b % p.pair_bs;
      srow0 = ((long long)(g2 * p.pair_bs + r2)) * p.L + (row0 - (long long)b * p.L);
    }
    if (pi > 0) {
      if (DBG == 1 && p.stop_after) {
        prime();  // (a dump left the ring in mid-stream)
      } else {    // the stream has wrapped: its tail copy of the first P pieces is in the ring
        wp = wsb + P * 1024;
        asm volatile("" : "+s"(wp));
      }
    }
    // ---------------- phase 0: X = attention output panel; acc = h + b_out1 ----------------
    x2barrier();  // previous panel: every wave is done with X and the chunk buffers
    {
      const int voff = (int)((lr * p.ldo + lg * 8) * 2);
#pragma unroll
      for (int i = 0; i < BM / 16 * KS / 8; ++i) {
        const int piece = 8 * i + wave;
        const int rbg = piece / KS, ks = piece - rbg * KS;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(o_srd, voff, (int)(((srow0 + rbg * 16) * p.ldo + ks * 32) * 2), 0);
        *reinterpret_cast<u32x4*>(smem + piece * 1024 + lane16) = v;
      }
    }
    load_rows_hres(h_srd, p.ldh, srow0);  // (lands behind the GEMM)
    acc_bias(0);
    x2barrier();
    ts(0);
    // ---------------- phase 1: h1 = attn @ Wo1^T + b + h;  X = LayerNorm2(h1) ----------------
    gemm5(smem + (4 * rg) * KS * 1024, X2_IC(KS));
    acc_add_hres();
    ts(1);
    round_to_hres();
    if (DBG == 1 && p.stop_after == 11) { dump_hres(row0); continue; }
    layernorm_to_x();
    acc_bias(1);  // q = LN2(h1) Wq^T: beta2 Wq^T (the folded LayerNorm shift) is the accumulators' initial value
    if (DBG == 1 && p.stop_after == 1) { dump_x(row0); continue; }
    x2barrier();
    ts(2);
    // ---------------- phase 2: q = LN2(h1) @ Wq^T -> X ----------------
    gemm5(smem + (4 * rg) * KS * 1024, X2_IC(KS));
    ts(3);
    x2barrier();  // all waves are done reading X
    store_x(one4);
    if (DBG == 1 && p.stop_after == 2) { dump_x(row0); continue; }
    x2barrier();
    ts(4);
    // ---------------- phase 3: text cross-attention, all heads, in place in X ----------------
    // unit = (16-row block, head): S^T = K q^T (6 key blocks x 2 d-steps), one-pass softmax over the lane's 24 keys and its 3
    // partner lanes, O^T = V^T P^T with P fed from the accumulator registers (the host arranges V^T's key order to match)
    {
      constexpr int NU = (BM / 16) * HEADS / 8;  // units per wave (5)
      vec8 kfr[12], vfr[12];
      auto unit = [&](int i, int& rbg, int& hd) __attribute__((always_inline)) {
        const int u = wave + 8 * i;   // head-major: the waves work on (nearly) the same head at the same time
        hd = u / (BM / 16);
        rbg = u - hd * (BM / 16);
      };
      auto load_k = [&](int hd) __attribute__((always_inline)) {
        const int so = (b * HEADS + hd) * 12 * 1024, vo = x2opq(lane16);
#pragma unroll
        for (int i = 0; i < 12; ++i)
          kfr[i] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(kf_srd, vo + (i & 3) * 1024, so + (i >> 2) * 4096, 0));
      };
      auto load_v = [&](int hd) __attribute__((always_inline)) {
        const int so = (b * HEADS + hd) * 12 * 1024, vo = x2opq(lane16);
#pragma unroll
        for (int i = 0; i < 12; ++i)
          vfr[i] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(vf_srd, vo + (i & 3) * 1024, so + (i >> 2) * 4096, 0));
      };
      int rbg, hd;
      unit(0, rbg, hd);
      load_k(hd);
      load_v(hd);
      for (int ui = 0; ui < NU; ++ui) {
        unit(ui, rbg, hd);
        int rbn, hdn;
        unit(ui + 1 < NU ? ui + 1 : ui, rbn, hdn);
        vec8 qf[2];
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) qf[ds] = *reinterpret_cast<const vec8*>(smem + (rbg * KS + 2 * hd) * 1024 + x2opq(lane16) + ds * 1024);
        f32x4 s[X2KB];
#pragma unroll
        for (int kb = 0; kb < X2KB; ++kb) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          s[kb] = T::mfma16(kfr[2 * kb], qf[0], z);
          s[kb] = T::mfma16(kfr[2 * kb + 1], qf[1], s[kb]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ui + 1 < NU) load_k(hdn);
        __builtin_amdgcn_sched_barrier(0);
        float mx = -1e30f;
#pragma unroll
        for (int kb = 0; kb < X2KB; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = kb * 16 + 4 * lg + e;
            const float sv = key < p.Lk ? s[kb][e] : -1e30f;
            s[kb][e] = sv;
            mx = fmaxf(mx, sv);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float neg_m = -mx * p.c;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < X2KB; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][e], p.c, neg_m));
            s[kb][e] = pv;
            psum += pv;
          }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        f32x4 o[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ss = 0; ss < 3; ++ss) {
          const uint4 pp = make_uint4(T::pack2(s[2 * ss][0], s[2 * ss][1]), T::pack2(s[2 * ss][2], s[2 * ss][3]),
                                      T::pack2(s[2 * ss + 1][0], s[2 * ss + 1][1]), T::pack2(s[2 * ss + 1][2], s[2 * ss + 1][3]));
          const vec8 pf = __builtin_bit_cast(vec8, pp);
#pragma unroll
          for (int db = 0; db < 4; ++db) o[db] = T::mfma16(vfr[db * 3 + ss], pf, o[db]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ui + 1 < NU) load_v(hdn);
        __builtin_amdgcn_sched_barrier(0);
        const float inv = 1.0f / psum;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          uint2 pk;
          pk.x = T::pack2(o[db][0] * inv, o[db][1] * inv);
          pk.y = T::pack2(o[db][2] * inv, o[db][3] * inv);
          *reinterpret_cast<uint2*>(smem + rbg * KS * 1024 + (4 * hd) * 512 + x2opq(lanew) + db * 512) = pk;
        }
      }
    }
    // acc = h1 + b_out2
    acc_bias(2);
    acc_add_hres();
    if (DBG == 1 && p.stop_after == 3) { dump_x(row0); continue; }
    x2barrier();
    ts(5);
    // ---------------- phase 4: h2 = a @ Wo2^T + b + h1;  X = LayerNorm3(h2) ----------------
    gemm5(smem + (4 * rg) * KS * 1024, X2_IC(KS));
    ts(6);
    round_to_hres();
    if (DBG == 1 && p.stop_after == 14) { dump_hres(row0); continue; }
    layernorm_to_x();
    acc_bias(3);  // acc = h2 + b_ff2
    acc_add_hres();
    load_rows_hres(x_srd, p.ldx, srow0);  // the block input (residual of proj_out): lands somewhere behind the feed-forward
    if (DBG == 1 && p.stop_after == 4) { dump_x(row0); continue; }
    x2barrier();
    ts(7);
    // ---------------- phase 5: GEGLU feed-forward ----------------
    // every wave: [projection of chunk c | barrier | GELU of chunk c, output projection of chunk c - 1 | barrier] — group 1 one
    // barrier behind group 0, so that on every SIMD one wave multiplies a projection while its partner does GELU arithmetic
    if (grp) x2barrier();
    f1(0);
    x2barrier();
    gelu(gb0);
    x2barrier();
    for (int c = 1; c < NCH; ++c) {
      f1(c);
      x2barrier();
      gelu(gb0 + (c & 1) * G::GB_BYTES);
      f2(gb0 + ((c - 1) & 1) * G::GB_BYTES);
      x2barrier();
    }
    x2barrier();
    f2(gb0 + ((NCH - 1) & 1) * G::GB_BYTES);
    x2barrier();
    if (!grp) x2barrier();
    ts(8);
    // ---------------- phase 6: h3 -> X; acc = x + b_po; out = h3 @ Wpo^T + ... ----------------
    store_x(one4);  // (every wave's last read of X — the last projection — was at least one barrier ago)
    if (DBG == 1 && p.stop_after == 5) { dump_x(row0); continue; }
    acc_bias(4);
    x2barrier();
    ts(9);
    gemm5(smem + (4 * rg) * KS * 1024, X2_IC(KS));
    acc_add_hres();
    ts(10);
    store_rows(row0);
    ts(11);
  }
  if (DBG == 2 && (wave & 3) == 0 && lane == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<u16*>(p.h)) + ((long long)blockIdx.x * 2 + grp) * 16;
    for (int i = 0; i < 12; ++i) dbg[i] = ta[i];
  }
#endif
}


// ===============================================================================================================
// xf2_head: x -> GroupNorm apply -> proj_in -> h (stored) -> LayerNorm1 -> q | k (stored [M, 2C]) and v^T (stored
// transposed per sample for the flash-attention kernel)
// ===============================================================================================================
template <typename T, int CC>
__global__ __launch_bounds__(X2NT) void xf2_head_kernel(const Xf2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using G = X2Cfg<CC>;
  using vec8 = typename T::vec8;
  constexpr int C = CC, CG = G::CG, KS = G::KS, BM = G::BM, P = G::P;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = CG == 4 ? wave >> 2 : 0, cg = CG == 4 ? wave & 3 : wave;
  const int lr = lane & 15, lg = lane >> 4;
  const int lane16 = lane * 16;
  const int lanew = (lg >> 1) * 256 + lr * 16 + (lg & 1) * 8;

  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int x0 = xcd * p.q;
  int xn = p.npanels - x0;
  xn = xn > p.q ? p.q : xn;
  const int nmine = xn > loc ? (xn - loc + p.gx - 1) / p.gx : 0;
  if (nmine == 0) return;

  constexpr int SPW = G::HEAD_PIECES + P;
  const __amdgpu_buffer_rsrc_t w_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wstream), 0, CG * SPW * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t xin_srd = x2srd(p.o, (((long long)p.M - 1) * p.ldo + C) * 2);
  const __amdgpu_buffer_rsrc_t h_srd = x2srd(p.h, (((long long)p.M - 1) * p.ldh + C) * 2);
  const __amdgpu_buffer_rsrc_t out_srd = x2srd(p.out, (((long long)p.M - 1) * p.ldout + 2 * C) * 2);
  const __amdgpu_buffer_rsrc_t prm_srd = x2srd(p.prm, 4 * C * 4);
  const long long nsamp = (long long)p.M / p.L;
  const __amdgpu_buffer_rsrc_t ab_srd = x2srd(p.ab, nsamp * 2 * C * 4);
  const __amdgpu_buffer_rsrc_t vt_srd = x2srd(p.vt, ((nsamp - 1) * p.vt_bs + (long long)(C - 1) * p.vt_ld + p.L) * 2);

  const int wsb = cg * (SPW * 1024);
  int wp = wsb;
  auto wload = [&](int j) __attribute__((always_inline)) -> vec8 {
    return __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(w_srd, x2opq(lane16) + (j & 3) * 1024, wp + (j >> 2) * 4096, 0));
  };
  vec8 wq[P];
  vec8 xa[4], xb[4];
  f32x4 acc[4][5];
#pragma unroll
  for (int i = 0; i < P; i += 2) {
    wq[i] = wload(0);
    wq[i + 1] = wload(1);
    X2_WP_ADV(2);
  }

  auto gemm5 = [&]() __attribute__((always_inline)) {
    const char* ap = smem + (4 * rg) * KS * 1024 + x2opq(lane16);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) xa[rb] = *reinterpret_cast<const vec8*>(ap + rb * KS * 1024);
    for (int k2 = 0; k2 < KS; k2 += 2) {
      x2_for<0, 2>([&](auto half_) __attribute__((always_inline)) {
        constexpr int half = decltype(half_)::value;
        if (half == 0 || k2 + 2 < KS) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) {
            if constexpr (half == 0) xb[rb] = *reinterpret_cast<const vec8*>(ap + rb * KS * 1024 + 1024);
            else xa[rb] = *reinterpret_cast<const vec8*>(ap + rb * KS * 1024 + 2048);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        x2_for<0, 5>([&](auto j_) __attribute__((always_inline)) {
          constexpr int j = decltype(j_)::value, slot = 5 * half + j;
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) acc[rb][j] = T::mfma16(wq[slot], half ? xb[rb] : xa[rb], acc[rb][j]);
          __builtin_amdgcn_sched_barrier(0);
          wq[slot] = wload(j);
          __builtin_amdgcn_sched_barrier(0);
        });
        X2_WP_ADV(5);
      });
      ap += 2048;
    }
  };
  auto acc_bias = [&](int prow) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prm_srd, x2opq(lg * 16) + 64 * j, (prow * C + 80 * cg) * 4, 0));
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][j] = b;
    }
  };
  auto store_rows = [&](__amdgpu_buffer_rsrc_t srd, long long ld, long long r0, int col0) __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int vo = x2opq((int)((((r0 + 64 * rg + 16 * rb + lr) * ld) + col0 + 80 * cg + 4 * lg) * 2));
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const u32x2 v = {T::pack2(acc[rb][j][0], acc[rb][j][1]), T::pack2(acc[rb][j][2], acc[rb][j][3])};
        __builtin_amdgcn_raw_buffer_store_b64(v, srd, vo + 32 * j, 0, 0);
      }
    }
  };

  for (int pi = 0; pi < nmine; ++pi) {
    const int panel = x0 + loc + pi * p.gx;
    const long long row0 = (long long)panel * BM;
    const int b = (int)(row0 / p.L);
    const int l0 = (int)(row0 - (long long)b * p.L);
    if (pi > 0) {  // the stream wraps: its tail copy of the first P pieces is in the ring
      wp = wsb + P * 1024;
      asm volatile("" : "+s"(wp));
    }
    // ---------------- phase 0: X = GroupNorm(x) = x * a[b, c] + s[b, c] (statistics precomputed) ----------------
    x2barrier();  // previous panel: every wave is done with X
    {
      const int voff = (int)((lr * p.ldo + lg * 8) * 2);
#pragma unroll
      for (int i = 0; i < BM / 16 * KS / 8; ++i) {
        const int piece = 8 * i + wave;
        const int rbg = piece / KS, ks = piece - rbg * KS;
        const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xin_srd, voff, (int)(((row0 + rbg * 16) * p.ldo + ks * 32) * 2), 0));
        const int ab_so = (b * 2 * C + ks * 32) * 4;
        const f32x4 a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ab_srd, lg * 32, ab_so, 0));
        const f32x4 a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ab_srd, lg * 32 + 16, ab_so, 0));
        const f32x4 s0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ab_srd, lg * 32, ab_so + C * 4, 0));
        const f32x4 s1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ab_srd, lg * 32 + 16, ab_so + C * 4, 0));
        float f[8];
        unpack8<T>(v, f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[e] = f[e] * a0[e] + s0[e];
          f[4 + e] = f[4 + e] * a1[e] + s1[e];
        }
        *reinterpret_cast<uint4*>(smem + piece * 1024 + lane16) = pack8<T>(f);
      }
    }
    acc_bias(0);
    x2barrier();
    // ---------------- phase 1: h = GN(x) @ Wpi^T + b -> HBM;  X = LayerNorm1(h) ----------------
    gemm5();
    store_rows(h_srd, p.ldh, row0, 0);
    {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[rb][j][e] = T::to_f32(T::from_f32(acc[rb][j][e]));
      float* red = reinterpret_cast<float*>(smem + G::X_BYTES);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) t += (acc[rb][j][0] + acc[rb][j][1]) + (acc[rb][j][2] + acc[rb][j][3]);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (lg == 0) red[cg * BM + 64 * rg + 16 * rb + lr] = t;
      }
      x2barrier();
      float mean[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < CG; ++w) m += red[w * BM + 64 * rg + 16 * rb + lr];
        mean[rb] = m * (1.0f / C);
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[rb][j][e] -= mean[rb];
            t += acc[rb][j][e] * acc[rb][j][e];
          }
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (lg == 0) red[(CG + cg) * BM + 64 * rg + 16 * rb + lr] = t;
      }
      x2barrier();
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < CG; ++w) v += red[(CG + w) * BM + 64 * rg + 16 * rb + lr];
        const float rstd = rsqrtf(v * (1.0f / C) + 1e-5f);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          uint2 pk;
          pk.x = T::pack2(acc[rb][j][0] * rstd, acc[rb][j][1] * rstd);
          pk.y = T::pack2(acc[rb][j][2] * rstd, acc[rb][j][3] * rstd);
          *reinterpret_cast<uint2*>(smem + (4 * rg + rb) * KS * 1024 + (5 * cg + j) * 512 + lanew) = pk;
        }
      }
    }
    // ---------------- phase 2 / 3: q, k (bias = the folded LayerNorm shift) -> out[:, 0:C], out[:, C:2C] ----------------
    acc_bias(1);
    x2barrier();
    gemm5();
    store_rows(out_srd, p.ldout, row0, 0);
    acc_bias(2);
    gemm5();
    store_rows(out_srd, p.ldout, row0, C);
    // ---------------- phase 4: v -> transposed through LDS -> v^T[b, c, l0 .. l0 + BM) ----------------
    acc_bias(3);
    gemm5();
    x2barrier();  // all waves are done reading X: reuse it as the [C channels][BM rows] transpose buffer
    {
      u16* ts = reinterpret_cast<u16*>(smem);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) ts[(80 * cg + 16 * j + 4 * lg + e) * BM + 64 * rg + 16 * rb + lr] = T::from_f32(acc[rb][j][e]);
      x2barrier();
      const int vt_so = (int)(((long long)b * p.vt_bs + l0) * 2);
      for (int q = tid; q < C * (BM / 8); q += X2NT) {
        const int n = q / (BM / 8), mc = q % (BM / 8);
        const u32x4 v = *reinterpret_cast<const u32x4*>(ts + n * BM + mc * 8);
        __builtin_amdgcn_raw_buffer_store_b128(v, vt_srd, (int)((n * p.vt_ld + mc * 8) * 2) + vt_so, 0, 0);
      }
    }
  }
#endif
}

int x2_grid(int npanels, int* q, int* gx) {
  *q = cdiv(npanels, 8);
  *gx = *q < 32 ? *q : 32;  // one workgroup per CU
  return 8 * *gx;
}

template <typename KT>
int x2_set_lds(KT kern, int bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int CC>
int x2_launch_tail(int dtype, int stop_after, int grid, hipStream_t s, const Xf2Params& p) {
  constexpr int LDS = X2Cfg<CC>::TAIL_LDS;
  static bool attr_set = false;
  if (!attr_set) {
    if (x2_set_lds(&xf2_tail_kernel<F16, 0, CC>, LDS) != 0 || x2_set_lds(&xf2_tail_kernel<BF16, 0, CC>, LDS) != 0 ||
        x2_set_lds(&xf2_tail_kernel<F16, 1, CC>, LDS) != 0 || x2_set_lds(&xf2_tail_kernel<BF16, 1, CC>, LDS) != 0) {
      dbir_set_error("dbir_xf_tail: cannot reserve %d bytes of LDS", LDS);
      return DBIR_ERR_LAUNCH;
    }
    attr_set = true;
  }
  if (stop_after == 99) {
#ifdef DBIR_DIAG
    if (x2_set_lds(&xf2_tail_kernel<F16, 2, CC>, LDS) != 0) return DBIR_ERR_LAUNCH;
    hipLaunchKernelGGL((xf2_tail_kernel<F16, 2, CC>), dim3(grid), dim3(X2NT), LDS, s, p);
#else
    dbir_set_error("dbir_xf_tail: stop_after 99 (section timing) needs a DBIR_DIAG build");
    return DBIR_ERR_ARG;
#endif
  } else if (stop_after) {  // debug instantiation (tests): intermediate dumps
    if (dtype == DBIR_F16) hipLaunchKernelGGL((xf2_tail_kernel<F16, 1, CC>), dim3(grid), dim3(X2NT), LDS, s, p);
    else hipLaunchKernelGGL((xf2_tail_kernel<BF16, 1, CC>), dim3(grid), dim3(X2NT), LDS, s, p);
  } else {
    if (dtype == DBIR_F16) hipLaunchKernelGGL((xf2_tail_kernel<F16, 0, CC>), dim3(grid), dim3(X2NT), LDS, s, p);
    else hipLaunchKernelGGL((xf2_tail_kernel<BF16, 0, CC>), dim3(grid), dim3(X2NT), LDS, s, p);
  }
  return DBIR_OK;
}

template <int CC>
int x2_launch_head(int dtype, int grid, hipStream_t s, const Xf2Params& p) {
  constexpr int LDS = X2Cfg<CC>::HEAD_LDS;
  static bool attr_set = false;
  if (!attr_set) {
    if (x2_set_lds(&xf2_head_kernel<F16, CC>, LDS) != 0 || x2_set_lds(&xf2_head_kernel<BF16, CC>, LDS) != 0) {
      dbir_set_error("dbir_xf_head: cannot reserve %d bytes of LDS", LDS);
      return DBIR_ERR_LAUNCH;
    }
    attr_set = true;
  }
  if (dtype == DBIR_F16) hipLaunchKernelGGL((xf2_head_kernel<F16, CC>), dim3(grid), dim3(X2NT), LDS, s, p);
  else hipLaunchKernelGGL((xf2_head_kernel<BF16, CC>), dim3(grid), dim3(X2NT), LDS, s, p);
  return DBIR_OK;
}

}  // namespace

// stream lengths of the second-generation kernels for inner width C: bytes of the head / tail weight streams (incl. one trailing KB
// that is never read: it keeps every length distinct from the first generation's) and floats of
// the tail's parameter block (5 bias rows + the feed-forward projection bias table)
extern "C" int dbir_xf2_geometry(int C, int* panel_rows, long long* head_bytes, long long* tail_bytes, int* tail_prm_floats) {
  DBIR_CHECK_ARG(C == 320 || C == 640, "dbir_xf2_geometry: the fused transformer kernels are built for C = 320 and 640 (got %d)", C);
  const bool a = C == 320;
  if (panel_rows) *panel_rows = a ? X2Cfg<320>::BM : X2Cfg<640>::BM;
  if (head_bytes) *head_bytes = a ? (long long)X2Cfg<320>::CG * (X2Cfg<320>::HEAD_PIECES + X2Cfg<320>::P) * 1024 + 1024
                                  : (long long)X2Cfg<640>::CG * (X2Cfg<640>::HEAD_PIECES + X2Cfg<640>::P) * 1024 + 1024;
  if (tail_bytes) *tail_bytes = a ? (long long)X2Cfg<320>::CG * (X2Cfg<320>::TAIL_PIECES + X2Cfg<320>::P) * 1024 + 1024
                                  : (long long)X2Cfg<640>::CG * (X2Cfg<640>::TAIL_PIECES + X2Cfg<640>::P) * 1024 + 1024;
  if (tail_prm_floats) *tail_prm_floats = 5 * C + (a ? X2Cfg<320>::B1_FLOATS : X2Cfg<640>::B1_FLOATS);
  return DBIR_OK;
}

// called by dbir_xf_tail (xformer.hip) when the weight stream has the second-generation length; arguments already validated
int dbir_xf2_tail_impl(int dtype, const void* attn_out, long long ldo, const void* h, long long ldh, const void* x, long long ldx,
                       void* out, long long ldout, int M, int L, int C, int pair_bs, const void* wstream, const float* prm,
                       const void* kfrag, const void* vfrag, int Lk, float scale, int stop_after, void* stream) {
  const int BM = C == 320 ? X2Cfg<320>::BM : X2Cfg<640>::BM;
  DBIR_CHECK_ARG(M > 0 && L > 0 && L % BM == 0 && M % L == 0, "dbir_xf_tail: M %d must be whole samples of L %d rows, L %% %d == 0", M, L, BM);
  DBIR_CHECK_ARG(stop_after == 0 || stop_after == 11 || stop_after == 14 || stop_after == 99 || (stop_after >= 1 && stop_after <= 5),
                 "dbir_xf_tail: stop_after %d is not a dump point of the second-generation kernel", stop_after);
  Xf2Params p;
  memset(&p, 0, sizeof(p));
  p.o = (const u16*)attn_out; p.ldo = ldo;
  p.h = (const u16*)h; p.ldh = ldh;
  p.x = (const u16*)x; p.ldx = ldx;
  p.out = (u16*)out; p.ldout = ldout;
  p.M = M; p.L = L; p.pair_bs = pair_bs;
  p.wstream = wstream; p.prm = prm;
  p.kf = (const u16*)kfrag; p.vf = (const u16*)vfrag;
  p.Lk = Lk; p.c = scale * 1.4426950408889634f;
  p.npanels = M / BM;
  p.stop_after = stop_after;
  const int grid = x2_grid(p.npanels, &p.q, &p.gx);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rc = C == 320 ? x2_launch_tail<320>(dtype, stop_after, grid, s, p) : x2_launch_tail<640>(dtype, stop_after, grid, s, p);
  if (rc != DBIR_OK) return rc;
  DBIR_CHECK_LAUNCH("dbir_xf_tail");
  return DBIR_OK;
}

int dbir_xf2_head_impl(int dtype, const void* x, long long ldx, const float* gn_scale_shift, void* h, long long ldh, void* qk,
                       long long ldqk, void* vt, long long vt_ld, long long vt_bstride, int M, int L, int C, const void* wstream,
                       const float* prm, void* stream) {
  const int BM = C == 320 ? X2Cfg<320>::BM : X2Cfg<640>::BM;
  DBIR_CHECK_ARG(M > 0 && L > 0 && L % BM == 0 && M % L == 0, "dbir_xf_head: M %d must be whole samples of L %d rows, L %% %d == 0", M, L, BM);
  Xf2Params p;
  memset(&p, 0, sizeof(p));
  p.o = (const u16*)x; p.ldo = ldx;
  p.h = (const u16*)h; p.ldh = ldh;
  p.out = (u16*)qk; p.ldout = ldqk;
  p.vt = (u16*)vt; p.vt_ld = vt_ld; p.vt_bs = vt_bstride;
  p.ab = gn_scale_shift;
  p.M = M; p.L = L;
  p.wstream = wstream; p.prm = prm;
  p.npanels = M / BM;
  const int grid = x2_grid(p.npanels, &p.q, &p.gx);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rc = C == 320 ? x2_launch_head<320>(dtype, grid, s, p) : x2_launch_head<640>(dtype, grid, s, p);
  if (rc != DBIR_OK) return rc;
  DBIR_CHECK_LAUNCH("dbir_xf_head");
  return DBIR_OK;
}
