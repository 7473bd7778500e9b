// Gate A of VERDICT r5 item 1, second form (round 6): the GEGLU feed-forward of a transformer block (reference
// diffbir/model/attention.py:19-45) as a standalone kernel, after tools/probes/xf_ff64.hip showed (profiles/r6_ff64_gateA_v1.txt)
// that (i) streaming the weights straight into registers sustains 1.5 - 1.9 PF/s of MFMA work and (ii) a one-wave-per-SIMD
// kernel cannot hide its own VALU work (the GELU arithmetic cost +31 ticks per 16-tick MFMA at C = 320).
//
//   out = GEGLU-FF(n3) + h2 + b2,   n3 = LayerNorm3(h2) rows (16 bit, normalised), W1 [8C, C] (values | gates), W2 [C, 4C]
//
// Design under test:
//   * EIGHT waves (two per SIMD), each owning 64 rows x 80 output columns as 4 x 5 blocks of v_mfma_f32_16x16x32: a
//     weight piece (16 columns x 32 k = 1 KB) feeds 4 MFMAs — the same weight bytes per FLOP as a 64 x 160 tile of
//     32x32x16 with HALF the accumulators (80 + 32 registers), so two waves fit a SIMD's register file;
//   * weights never touch LDS: every wave streams its own pieces (C = 640: private; C = 320: shared by the two row groups)
//     into a 10-piece register ring with 16-byte buffer loads, consumption order = stream order, one running scalar offset;
//   * LDS holds only the activation panel image (80 KB, [16-row block][k-step of 32][lane][16 B]) and the double-buffered
//     GEGLU chunk (2 x 16 KB);
//   * the two wave groups (waves 0-3 / 4-7 = one wave of each on every SIMD) run ONE BARRIER APART: while one group's waves
//     multiply the GEGLU projection of a chunk (MFMA-dense), their SIMD partners do the GELU arithmetic of the previous one
//     (VALU-dense) and its output projection — one workgroup barrier per 120 / 240 MFMAs per wave.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/probes/xf_ff16.hip -o /tmp/xf_ff16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>
#include <vector>

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned short u16;

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

template <int CC>
struct HCfg {
  static constexpr int C = CC, CG = CC / 80, RG = 8 / CG, BM = 64 * RG, KS = CC / 32;
  static constexpr int CHH = 16 * CG;                  // hidden units per chunk: 16 per column group (64 / 128)
  static constexpr int NCH = 4 * CC / CHH;             // chunks (20)
  static constexpr int GK = CHH / 32;                  // k-steps of a chunk's output projection (2 / 4)
  static constexpr int F1P = 2 * KS, F2P = 5 * GK;     // pieces of a projection / output-projection sub-block (20 / 40, 10 / 20)
  static constexpr int P = 10;                         // weight ring: pieces in flight per wave (divides F1P and F2P)
  static constexpr int SPW = NCH * (F1P + F2P) + P;    // pieces per column-group stream (+ a copy of its first P)
  static constexpr int X_BYTES = BM * CC * 2;          // 81920
  static constexpr int GB_BYTES = BM * CHH * 2;        // 16384
  static constexpr int B1_BYTES = NCH * CG * 32 * 4;   // 10240 / 20480: the projection bias of every chunk, [chunk][column group][value | gate][16]
  static constexpr int LDS = X_BYTES + 2 * GB_BYTES + B1_BYTES;
};

struct FfParams {
  const u16* n3; const u16* h2; u16* out;   // [M, C] row-major
  const void* wstream;                      // [CG][SPW] pieces of 1 KB, consumption order (see run())
  const float* b1;                          // [NCH][CG][value | gate][16] f32
  const float* b2;                          // [C]
  unsigned long long* ticks;                // [grid][8 waves][2]: s_memtime ticks in the chunk loops / in all
  int M, npanels, q, gx;
};

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float h2f(u16 v) { return (float)__builtin_bit_cast(f16, v); }
// gelu(x) = x * Phi(x) ~ x / (1 + 2^(x * p(min(x^2, 64)))), p = -log2(e) * (1.5961 + 0.07331 x^2 - 0.000582 x^4):
// max abs error 8.1e-5 against the erf form (0.17 f16 ulp at 1)
__device__ __forceinline__ float gelu_sp(float x) {
  const float x2 = fminf(x * x, 64.0f);
  float p = __builtin_fmaf(8.39458781e-04f, x2, -1.05767970e-01f);
  p = __builtin_fmaf(p, x2, -2.30265908f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * x));
}
__device__ __forceinline__ void xbarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CC, int ABL>  // ABL (timing only): 1 = no weight loads in the loops, 2 = no GELU arithmetic, 4 = no LDS fragment reads in the loops, 8 = no barriers in the loops
__global__ __launch_bounds__(512) void ff16_kernel(const FfParams p) {
  using G = HCfg<CC>;
  constexpr int C = CC, CG = G::CG, KS = G::KS, GK = G::GK, NCH = G::NCH, BM = G::BM, P = G::P;
  static_assert(G::F1P % P == 0 && G::F2P % P == 0, "every sub-block must start at ring slot 0");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                   // wave group: one wave of each group on every SIMD
  const int rg = CG == 4 ? wave >> 2 : 0, cg = CG == 4 ? wave & 3 : wave;
  const int lr = lane & 15, lg = lane >> 4;
  const int lane16 = lane * 16;

  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int x0 = xcd * p.q;
  int xn = p.npanels - x0;
  xn = xn > p.q ? p.q : xn;
  const int nmine = xn > loc ? (xn - loc + p.gx - 1) / p.gx : 0;
  if (nmine == 0) return;

  const __amdgpu_buffer_rsrc_t w_srd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wstream), 0, CG * G::SPW * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t b1_srd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b1), 0, NCH * CG * 32 * 4, 0x00020000);
  const long long abytes = (long long)p.M * C * 2;
  const __amdgpu_buffer_rsrc_t n3_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.n3), 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t h2_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.h2), 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t out_srd = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b2_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b2), 0, C * 4, 0x00020000);
  // a column group's stream is consumed strictly in order: ONE running scalar offset, P pieces ahead of the consumer
  const int wsb = cg * (G::SPW * 1024);
  int wp = wsb;
  auto wload = [&](int j) __attribute__((always_inline)) -> f16x8 {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(w_srd, lane16 + j * 1024, wp, 0));
  };
#define WP_ADV(N) do { wp += (N) * 1024; asm volatile("" : "+s"(wp)); } while (0)
#define IC(N) std::integral_constant<int, (N)>{}
  auto for_range = [&](auto lo_, auto hi_, auto&& fn) __attribute__((always_inline)) {
    constexpr int lo = decltype(lo_)::value, hh = decltype(hi_)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) { (fn(std::integral_constant<int, lo + I>{}), ...); }
    (std::make_integer_sequence<int, hh - lo>{});
  };

  unsigned long long t_loop = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  f16x8 wq[P];
  f16x8 xa[4], xb[4];
  f32x4 acc[4][5];    // [16-row block][16-column block]: the output rows
  f32x4 gacc[4][2];   // [16-row block][value | gate]: the GEGLU projection of the chunk in flight

#pragma unroll
  for (int i = 0; i < P; i += 2) {
    wq[i] = wload(0);
    wq[i + 1] = wload(1);
    WP_ADV(2);
  }
  char* const gb0 = smem + G::X_BYTES;
  {  // projection bias -> LDS, once per workgroup
    float* bl = reinterpret_cast<float*>(smem + G::X_BYTES + 2 * G::GB_BYTES);
    for (int i = tid; i < G::B1_BYTES / 16; i += 512)
      reinterpret_cast<f32x4*>(bl)[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b1_srd, i * 16, 0, 0));
  }

  // ---- GEGLU projection of chunk c: K = C, 4 row blocks x (value, gate)
  auto f1 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
      xa[rb] = *reinterpret_cast<const f16x8*>(smem + ((4 * rg + rb) * KS + 0) * 1024 + lane16);
    for_range(IC(0), IC(KS), [&](auto ks_) __attribute__((always_inline)) {
      constexpr int ks = decltype(ks_)::value;
      // fragments of the next k-step into the other register set (the projection wave is the critical path of an interval)
      if constexpr (ks + 1 < KS && !(ABL & 4)) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          if constexpr (ks & 1) xa[rb] = *reinterpret_cast<const f16x8*>(smem + ((4 * rg + rb) * KS + ks + 1) * 1024 + lane16);
          else xb[rb] = *reinterpret_cast<const f16x8*>(smem + ((4 * rg + rb) * KS + ks + 1) * 1024 + lane16);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      for_range(IC(0), IC(2), [&](auto nb_) __attribute__((always_inline)) {
        constexpr int nb = decltype(nb_)::value, slot = (2 * ks + nb) % P;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const f16x8 xf = (ks & 1) ? xb[rb] : xa[rb];
          if constexpr (ks == 0) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            gacc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[slot], xf, z, 0, 0, 0);
          } else {
            gacc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[slot], xf, gacc[rb][nb], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 1)) wq[slot] = wload(nb);
        __builtin_amdgcn_sched_barrier(0);
      });
      WP_ADV(2);
    });
  };
  // ---- g = (value + b) * gelu(gate + b) of the chunk whose projection this wave has just finished -> chunk image gbw
  auto gelu = [&](int c, char* gbw) __attribute__((always_inline)) {
    const f32x4* bl = reinterpret_cast<const f32x4*>(smem + G::X_BYTES + 2 * G::GB_BYTES + ((c * CG + cg) * 32) * 4) + lg;
    const f32x4 bq[2] = {bl[0], bl[4]};
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (ABL & 2) v[e] = gacc[rb][0][e] + gacc[rb][1][e];
        else v[e] = (gacc[rb][0][e] + bq[0][e]) * gelu_sp(gacc[rb][1][e] + bq[1][e]);
      }
      uint2 pk;
      pk.x = pack2(v[0], v[1]);
      pk.y = pack2(v[2], v[3]);
      // hidden units 16 cg + 4 lg .. + 3 of row 16 (4 rg + rb) + lr: piece (row block, cg / 2), lane slot (2 (cg & 1) + lg / 2) * 16 + lr, half lg & 1
      const int off = ((4 * rg + rb) * GK + (cg >> 1)) * 1024 + ((2 * (cg & 1) + (lg >> 1)) * 16 + lr) * 16 + (lg & 1) * 8;
      *reinterpret_cast<uint2*>(gbw + off) = pk;
    }
  };
  // ---- output projection of one chunk: K = chunk, 4 row blocks x 5 column blocks
  auto f2 = [&](const char* gbr) __attribute__((always_inline)) {
    for_range(IC(0), IC(GK), [&](auto k_) __attribute__((always_inline)) {
      constexpr int k = decltype(k_)::value;
#pragma unroll
      for (int rb = 0; rb < ((ABL & 4) ? 0 : 4); ++rb)
        xa[rb] = *reinterpret_cast<const f16x8*>(gbr + ((4 * rg + rb) * GK + k) * 1024 + lane16);
      __builtin_amdgcn_sched_barrier(0);
      for_range(IC(0), IC(5), [&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value, slot = (5 * k + j) % P;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[slot], xa[rb], acc[rb][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 1)) wq[slot] = wload(j);
        __builtin_amdgcn_sched_barrier(0);
      });
      WP_ADV(5);
    });
  };

  for (int pi = 0; pi < nmine; ++pi) {
    const int panel = x0 + loc + pi * p.gx;
    const long long row0 = (long long)panel * BM;
    xbarrier();  // previous panel: every wave is done with X and the chunk buffers
    wp = wsb + P * 1024;
    asm volatile("" : "+s"(wp));
    // panel -> X image [16-row block][k-step of 32][lane][16 B] (through registers; the product kernel overlaps this)
    {
      constexpr int NP = BM / 16 * KS;  // 80 pieces
      const int voff = (int)((lr * C + lg * 8) * 2);
#pragma unroll
      for (int i = 0; i < NP / 8; ++i) {
        const int piece = 8 * i + wave;
        const int rb = piece / KS, ks = piece - rb * KS;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(n3_srd, voff, (int)(((row0 + rb * 16) * C + ks * 32) * 2), 0);
        *reinterpret_cast<u32x4*>(smem + piece * 1024 + lane16) = v;
      }
    }
    // acc = h2 + b2: a lane owns 4 consecutive columns (80 cg + 16 j + 4 lg ..) of row 64 rg + 16 rb + lr
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int col = 80 * cg + 16 * j + 4 * lg;
        const int row = 64 * rg + 16 * rb + lr;
        const auto hv = __builtin_amdgcn_raw_buffer_load_b64(h2_srd, (int)((row * C + col) * 2), (int)(row0 * C * 2), 0);
        const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b2_srd, col * 4, 0, 0));
        acc[rb][j][0] = h2f((u16)(hv[0] & 0xffff)) + bv[0];
        acc[rb][j][1] = h2f((u16)(hv[0] >> 16)) + bv[1];
        acc[rb][j][2] = h2f((u16)(hv[1] & 0xffff)) + bv[2];
        acc[rb][j][3] = h2f((u16)(hv[1] >> 16)) + bv[3];
      }
    xbarrier();
    // every wave: [projection of chunk c | barrier | GELU of chunk c, output projection of chunk c - 1 | barrier] — group 1 one barrier
    // behind group 0, so that on every SIMD one wave multiplies a projection while its partner does GELU arithmetic
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define LB() do { if constexpr (!(ABL & 8)) xbarrier(); } while (0)
    if (grp) LB();   // group 1 runs one barrier behind group 0
    f1();
    LB();
    gelu(0, gb0);
    LB();
    for (int c = 1; c < NCH; ++c) {
      f1();
      LB();
      gelu(c, gb0 + (c & 1) * G::GB_BYTES);
      f2(gb0 + ((c - 1) & 1) * G::GB_BYTES);
      LB();
    }
    LB();
    f2(gb0 + ((NCH - 1) & 1) * G::GB_BYTES);
    LB();
    if (!grp) LB();
    t_loop += __builtin_amdgcn_s_memtime() - t0;
    // ---------------- rows out: 4 consecutive columns per lane -> 8-byte stores ----------------
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int col = 80 * cg + 16 * j + 4 * lg;
        const long long row = row0 + 64 * rg + 16 * rb + lr;
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
        const u32x2 v = {pack2(acc[rb][j][0], acc[rb][j][1]), pack2(acc[rb][j][2], acc[rb][j][3])};
        __builtin_amdgcn_raw_buffer_store_b64(v, out_srd, (int)((row * C + col) * 2), 0, 0);
      }
  }
  if (p.ticks && lane == 0) {
    p.ticks[(blockIdx.x * 8 + wave) * 2] = t_loop;
    p.ticks[(blockIdx.x * 8 + wave) * 2 + 1] = __builtin_amdgcn_s_memtime() - t_begin;
  }
}

// ------------------------------------------------------------------------------------------------------------------
static float frand(uint64_t& s) {  // uniform (-1, 1)
  s = s * 6364136223846793005ULL + 1442695040888963407ULL;
  return (float)((s >> 33) & 0xffffff) / 8388608.0f - 1.0f;
}
static float nrand(uint64_t& s) { return (frand(s) + frand(s) + frand(s) + frand(s)) * 0.8660254f; }
static u16 f2h(float v) { f16 h = (f16)v; u16 r; memcpy(&r, &h, 2); return r; }
static float h2fh(u16 v) { f16 h; memcpy(&h, &v, 2); return (float)h; }

// [N, K] -> 16 x 32 piece: lane 16 lg + lr = W[row0 + lr, 32 ks + 8 lg .. + 8]
static void put_piece16(u16* dst, const std::vector<u16>& w, int K, int row0, int ks) {
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 8; ++e) dst[l * 8 + e] = w[(size_t)(row0 + (l & 15)) * K + 32 * ks + 8 * (l >> 4) + e];
}

template <int CC, int ABL = 0>
static void run(int M, int reps) {
  using G = HCfg<CC>;
  constexpr int C = CC, H = 4 * CC;
  uint64_t seed = 1234567 + CC;
  std::vector<u16> n3((size_t)M * C), h2((size_t)M * C), w1((size_t)2 * H * C), w2((size_t)C * H);
  std::vector<float> b1(2 * H), b2(C);
  for (auto& v : n3) v = f2h(nrand(seed));
  for (auto& v : h2) v = f2h(nrand(seed));
  for (auto& v : w1) v = f2h(nrand(seed) / sqrtf((float)C));
  for (auto& v : w2) v = f2h(nrand(seed) / sqrtf((float)H));
  for (auto& v : b1) v = 0.1f * nrand(seed);
  for (auto& v : b2) v = 0.1f * nrand(seed);
  // streams, consumption order per column group: F1(0) [F1(c) F2(c-1)] c = 1..NCH-1, F2(NCH-1), + copy of the first P
  std::vector<u16> ws((size_t)G::CG * G::SPW * 512);
  std::vector<float> b1p((size_t)G::NCH * G::CG * 32);
  for (int cg = 0; cg < G::CG; ++cg) {
    u16* dst = ws.data() + (size_t)cg * G::SPW * 512;
    auto put_f1 = [&](int c) {
      const int h0 = c * G::CHH + 16 * cg;
      for (int ks = 0; ks < G::KS; ++ks)
        for (int nb = 0; nb < 2; ++nb) { put_piece16(dst, w1, C, nb ? H + h0 : h0, ks); dst += 512; }
    };
    auto put_f2 = [&](int c) {
      for (int k = 0; k < G::GK; ++k)
        for (int j = 0; j < 5; ++j) { put_piece16(dst, w2, H, 80 * cg + 16 * j, c * G::GK + k); dst += 512; }
    };
    put_f1(0);
    for (int c = 1; c < G::NCH; ++c) { put_f1(c); put_f2(c - 1); }
    put_f2(G::NCH - 1);
    memcpy(dst, ws.data() + (size_t)cg * G::SPW * 512, (size_t)G::P * 1024);
    for (int c = 0; c < G::NCH; ++c)
      for (int nb = 0; nb < 2; ++nb)
        for (int i = 0; i < 16; ++i) b1p[(size_t)(c * G::CG + cg) * 32 + nb * 16 + i] = b1[(nb ? H : 0) + c * G::CHH + 16 * cg + i];
  }
  u16 *d_n3, *d_h2, *d_out, *d_ws;
  float *d_b1, *d_b2;
  unsigned long long* d_ticks;
  CK(hipMalloc(&d_ticks, 256 * 8 * 2 * 8));
  CK(hipMalloc(&d_n3, n3.size() * 2)); CK(hipMalloc(&d_h2, h2.size() * 2)); CK(hipMalloc(&d_out, n3.size() * 2));
  CK(hipMalloc(&d_ws, ws.size() * 2)); CK(hipMalloc(&d_b1, b1p.size() * 4)); CK(hipMalloc(&d_b2, b2.size() * 4));
  CK(hipMemcpy(d_n3, n3.data(), n3.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_h2, h2.data(), h2.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ws, ws.data(), ws.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b1, b1p.data(), b1p.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b2, b2.data(), b2.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_out, 0, n3.size() * 2));
  FfParams p;
  p.n3 = d_n3; p.h2 = d_h2; p.out = d_out; p.wstream = d_ws; p.b1 = d_b1; p.b2 = d_b2; p.M = M; p.ticks = d_ticks;
  p.npanels = M / G::BM;
  p.q = (p.npanels + 7) / 8;
  p.gx = p.q < 32 ? p.q : 32;
  const int grid = 8 * p.gx;
  auto kern = ff16_kernel<CC, ABL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, sum = 0.f;
  for (int r = 0; r < reps + 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::LDS, 0, p);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
  }
  CK(hipGetLastError());
  const double fl = 24.0 * M * C * C;
  std::vector<unsigned long long> ticks((size_t)grid * 16);
  CK(hipMemcpy(ticks.data(), d_ticks, ticks.size() * 8, hipMemcpyDeviceToHost));
  double tl = 0, ta = 0;
  for (int i = 0; i < grid * 8; ++i) { tl += ticks[2 * i]; ta += ticks[2 * i + 1]; }
  tl /= grid * 8; ta /= grid * 8;
  const double mf = fl / 2 / 8192 / (grid * 8.0);  // 16x16x32 MFMAs per wave
  printf("ff16 C%d abl %d M %d grid %d: min %.1f us (%.0f TF/s)  mean %.1f us (%.0f TF/s) | ticks per wave: loops %.0f of %.0f (%.0f%%), %.2f per MFMA (2 waves share a pipe)\n",
         C, ABL, M, grid, best * 1e3, fl / best / 1e9, sum / reps * 1e3, fl / (sum / reps) / 1e9, tl, ta, 100 * tl / ta, tl / mf);
  // check a sample of rows against a host reference (f32 accumulation, exact erf GELU, g rounded to 16 bit as the kernel does)
  std::vector<u16> out((size_t)M * C);
  CK(hipMemcpy(out.data(), d_out, out.size() * 2, hipMemcpyDeviceToHost));
  double max_err = 0, max_ref = 0;
  const int rows[] = {0, 1, 15, 16, 31, 32, 63, 64, 65, 127, 128, 200, M / 2 + 77, M - 1};
  std::vector<float> g(H);
  for (int row : rows) {
    if (row >= M || ABL) continue;
    for (int h = 0; h < H; ++h) {
      float v = b1[h], gt = b1[H + h];
      for (int k = 0; k < C; ++k) {
        const float x = h2fh(n3[(size_t)row * C + k]);
        v += x * h2fh(w1[(size_t)h * C + k]);
        gt += x * h2fh(w1[(size_t)(H + h) * C + k]);
      }
      const float ge = 0.5f * gt * (1.0f + erff(gt * 0.70710678f));
      g[h] = h2fh(f2h(v * ge));
    }
    for (int n = 0; n < C; ++n) {
      float a = b2[n] + h2fh(h2[(size_t)row * C + n]);
      for (int h = 0; h < H; ++h) a += g[h] * h2fh(w2[(size_t)n * H + h]);
      const double err = fabs((double)a - (double)h2fh(out[(size_t)row * C + n]));
      if (getenv("FF_DEBUG") && err > 0.05 && (n % 16) < 2) printf("      row %d col %d: ref %.4f got %.4f\n", row, n, a, h2fh(out[(size_t)row * C + n]));
      max_err = err > max_err ? err : max_err;
      max_ref = fabs(a) > max_ref ? fabs(a) : max_ref;
    }
  }
  if (!ABL) printf("   check: max abs err %.4g (max |ref| %.3g) %s\n", max_err, max_ref, max_err < 2e-2 * (max_ref > 1 ? max_ref : 1) ? "OK" : "MISMATCH");
  CK(hipFree(d_n3)); CK(hipFree(d_h2)); CK(hipFree(d_out)); CK(hipFree(d_ws)); CK(hipFree(d_b1)); CK(hipFree(d_b2));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int small = argc > 2 ? atoi(argv[2]) : 0;
  run<320>(small ? 4096 : 65536, reps);
  run<640>(small ? 1024 : 16384, reps);
  if (small) return 0;
  run<320, 3>(65536, reps); run<320, 7>(65536, reps); run<320, 11>(65536, reps); run<320, 15>(65536, reps);
  run<640, 3>(16384, reps); run<640, 7>(16384, reps); run<640, 11>(16384, reps); run<640, 15>(16384, reps);
  run<320>(32768, reps);
  run<640>(8192, reps);
  return 0;
}
