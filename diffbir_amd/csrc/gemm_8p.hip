// Implicit-GEMM (3x3 convolution / 1x1 / linear) on gfx950 MFMA: 256 x 320 output tile, fine-phase schedule.
//
// Why a new kernel: the direct-to-LDS / halo-patch kernels spend one or two workgroup
// barriers per K tile with every wave doing the same thing at the same time (fragment reads, then MFMAs): the matrix time
// is ADDITIVE to the LDS-read / staging skeleton (profiles/r2_halo_ablation.txt: 132 us full, 82 us without any MFMA), and
// their 32 x 160 wave tile needs 6 ds_read_b128 per 5 MFMAs (~150 B/clk of the LDS's 256).  This kernel follows the
// 8-phase template of cdna_hip_programming.md §5 (T3+T4+T5) re-derived for the UNet's N = 320 k channel counts:
//
//   * 512 threads = 8 waves as 4 (M) x 2 (N); a wave owns 64 x 160 outputs = 2 x 5 accumulators of v_mfma_f32_32x32x16
//     (160 registers): 7 ds_read_b128 per 10 MFMAs (~88 B/clk at full matrix rate), 256 x 320 per workgroup — one tile
//     per CU for the 64x64-level convolutions (65536 x 320), staged bytes per MFMA 0.7x those of a 256 x 160 tile.
//   * K advances in STAGES of 32 (one tap, 32 channels): LDS rows of 64 B, 4-slot ring (4 x 36 KB).  A stage is consumed
//     in TWO PHASES (one 16-deep MFMA step each).  Phase = { 7 fragment reads + this phase's share of the direct-to-LDS
//     copies | s_barrier | 10 MFMAs under s_setprio 1 | s_barrier }.  The two wave groups (waves 0-3 / 4-7: one wave of each
//     per SIMD) run ONE BARRIER apart, so on every SIMD one wave is in its MFMA section while its partner reads
//     fragments and issues copies: the matrix pipe sees back-to-back MFMAs, the LDS / copy issue happens in its shadow.
//   * Copies run 2-3 stages ahead with counted vmcnt, never drained in the loop:
//        phase (s, 0): weight rows of stage s + 2      phase (s, 1): activation rows of stage s + 3, then
//        s_waitcnt vmcnt(N) for "stage s + 1 landed" (N = the copies issued after it: 6 / 7 per wave).
//     RAW: the wait sits in front of the first barrier of phase (s, 1); the first read of stage s + 1 is in phase (s + 1, 0),
//          i.e. behind that barrier for both wave groups (guide: "read one phase after the wait that retires it").
//     WAR: activation rows of stage s + 3 refill the slot of stage s - 1, last read in phase (s - 1, 1) — two phases
//          earlier, retired by the lgkmcnt wait in front of that phase's MFMAs; weight rows of stage s + 2 refill the
//          slot of stage s - 2.
//   * Operand fetch as in gemm_glds.hip: two block-local buffer descriptors, per-lane 32-bit offsets computed once, tap /
//     channel-slice part in the scalar offset, hardware zero fill for padding; LDS image lane-linear with the bank swizzle
//     (chunk ^ ((row >> 2) & 3)) on the SOURCE side and on the ds_read_b128 (guide rule 21): conflict-free.
//   * Split-K INSIDE the launch (the 32x32 / 16x16 levels have only 128 / 64 tiles): the slices of a tile are adjacent LOGICAL
//     ids, which the bijective XCD remap keeps on one XCD only when (tiles * splitk) / 8 is a multiple of splitk — in general
//     a tile's slices straddle two XCDs (e.g. splitk 3 / 9), and correctness never depends on the placement: every slice
//     writes its f32 accumulators, in register order, as WRITE-THROUGH (sc1) 16-byte stores, every wave drains them
//     (s_waitcnt vmcnt(0)) in front of the workgroup barrier, one lane takes an agent-scope ticket; the last arriver reads
//     the slabs with sc1 loads (they bypass its L1, and an sc1-stored line is not kept in any XCD's L2, so the bytes come
//     from memory) and sums them in slice order.  This is the drained-write-through hand-off of cdna_hip_programming.md §6
//     Guideline 16 (R1) / MI355X_MICROARCH.md "Valid forms": "sc1 loads may replace the acquire only when the producer
//     stored sc1" — no buffer_wbl2 / buffer_inv is needed, and a release fence here would only add an L2 write-back of
//     unrelated dirty lines (1.7 - 6.5 us per workgroup).  No spin, no co-residency requirement, deterministic.
//   * PH4 (round 5, desc.upsample == 2): nearest-x2 upsample + 3x3 convolution collapsed into FOUR 2x2 convolutions on the
//     low-resolution input, one per output parity (a, b): output pixel (2i + a, 2j + b) reads the 2x2 low-resolution
//     neighbourhood (i + a - 1 .., j + b - 1 ..) against the 3x3 taps that fall on each of those pixels SUMMED on the host
//     (ops.pack_conv3x3_up4): 4 / 9 of the multiplies, exact algebra.  One launch runs all four phases (grid = 4 x the
//     low-resolution tiles, phase-major), the epilogue scatters a tile's rows to their output pixels.
//   * Epilogue: the math of gemm_epilogue.h (bias, time-embedding row vector, SiLU / GELU / LeakyReLU, scale, residual,
//     GroupNorm column sums of the stored values) in two row passes of 128 rows (the 16-bit tile of 256 x 320 does not fit
//     the LDS at once).
#include <stdlib.h>

#include <type_traits>

#ifdef DBIR_DIAG  // `DBIR_DIAG=1 sh build.sh`: diagnostic instantiations of the kernel selected by env DBIR_P8_VAR
#define P8_VARIANTS 1  // (tools/probes/p8_diag.py: section stamps, ablations, schedule variants — never in the production library)
#endif

#include "common.h"
#include "gemm_epilogue.h"

extern thread_local int g_dbir_stats_rows;  // gemm.hip

namespace {

struct P8 {
  dbir_gemm_desc d;
  int Hv, Wv;   // virtual (upsampled) input extent for conv bounds checks
  int nkc;      // 32-channel slices per tap (conv) / K / 32 (linear)
  int ntaps;    // 9 / 1
  int mtiles, ntiles;
  int vec_bias, vec_rv;
  int splitk, st_per;  // K slices (1 = off), stages per slice
  float* ws;           // split-K: f32 accumulator slabs [tile][slice][40][512] float4
  unsigned* cnt;       // split-K: arrival counters [tile], zeroed ahead of the launch
  long long a_elems;
  int ph_tiles;        // PH4: tiles per output parity (mtiles * ntiles); 0 otherwise
  int st_tps;          // PH4: 256-row tiles per sample and parity (statistics rows are grouped per sample), 0 = no statistics
  int lgW, lgHW;       // PH4: log2(Wi), log2(Hi * Wi)
};

typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm8() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int kBM = 256, kBN = 320;

template <typename T, int VAR>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const P8 p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WN = 2, MI = 2, NJ = 5;
  constexpr int BM = kBM, BN = kBN;
  constexpr int ROWB = 64;                        // bytes per LDS tile row (32 halfs)
  constexpr int A_BYTES = BM * ROWB;              // 16 KB
  constexpr int SLOT = (BM + BN) * ROWB;          // 36 KB
  constexpr int RND = 512 * 16;                   // bytes per staging round of the whole block (128 rows)
  constexpr int LA = 2;                           // activation rounds per stage (every wave)
  constexpr bool UPS = (VAR & 1) != 0;            // nearest-x2 upsample folded into the activation gather
  // diagnostic variants (P8_VARIANTS builds only; timing, results meaningless for 4 / 8 / 16)
  constexpr bool NOPRIO = (VAR & 2) != 0, NOMFMA = (VAR & 4) != 0, NOSTAGE = (VAR & 8) != 0, NOREAD = (VAR & 16) != 0;
  constexpr bool TIMING = (VAR & 32) != 0, LGK_EARLY = (VAR & 64) != 0;
  constexpr bool ONEBAR = (VAR & 128) != 0;       // one-barrier-per-phase schedule (see the ONEBAR block of the main loop)
  constexpr bool PH4 = (VAR & 256) != 0;          // parity-collapsed nearest-x2 upsample convolution (2x2 taps, 4 phases)
  static_assert(!(PH4 && UPS), "PH4 replaces the upsampled gather");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned long long rt_kernel = TIMING ? __builtin_amdgcn_s_memrealtime() : 0;
  const dbir_gemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lq = lane & 31, hi = lane >> 5;
  const int grp = wave >> 2;                      // wave group: 0 = waves 0-3 (also holds the third weight round)

  // ---- XCD-aware tile mapping (bijective); the K slices of a tile are consecutive logical ids -> same XCD ----
  int tm, tn, ksp, tile, phase = 0;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    tile = lid / p.splitk;
    ksp = lid - tile * p.splitk;
    int t2 = tile;
    if constexpr (PH4) {
      phase = tile / p.ph_tiles;
      t2 = tile - phase * p.ph_tiles;
    }
    tn = t2 % p.ntiles;
    tm = t2 / p.ntiles;
  }
  const int pa = phase >> 1, pb = phase & 1;          // PH4: output row / column parity of this tile
  const int M = PH4 ? d.B * d.Hi * d.Wi : d.M;        // rows of the tile space (PH4: low-resolution pixels)
  const int rW = PH4 ? d.Wi : d.Wo, rHW = PH4 ? d.Hi * d.Wi : d.Ho * d.Wo;
  const int py = PH4 ? 1 - pa : d.pad, px = PH4 ? 1 - pb : d.pad;   // tap (0, 0) sits at (oy * stride - py, ox * stride - px)
  const u16* __restrict__ Ag = reinterpret_cast<const u16*>(d.A);
  const u16* __restrict__ Wg = reinterpret_cast<const u16*>(d.W);
  const bool conv = d.mode == DBIR_MODE_CONV3X3;

  // ---- staging roles: thread -> LDS chunk position (row = (tid >> 2) + 128 * round, cpos = tid & 3) ----
  const int srow = tid >> 2;
  const int skey = (srow >> 2) & 3;
  const int cch = ((tid & 3) ^ skey) * 8;         // logical K offset (halfs) of the chunk this thread fetches

  constexpr int OOB = 0x7fffff00;
  long long a_ref;
  {
    const int m0 = tm * BM < M ? tm * BM : M - 1;
    if (conv) {
      const int hw = rHW;
      const int b0 = m0 / hw, rem0 = m0 - b0 * hw;
      const int oy0 = rem0 / rW;
      int sy0 = oy0 * d.stride - py;
      if (UPS) sy0 >>= 1;
      a_ref = ((long long)b0 * d.Hi * d.Wi + (long long)sy0 * d.Wi - 2) * d.Cin;
    } else {
      a_ref = (long long)m0 * d.lda;
    }
  }
  long long a_left = (p.a_elems - a_ref) * 2;
  if (a_left > 0x7ffffe00LL) a_left = 0x7ffffe00LL;
  if (a_left < 0) a_left = 0;
  const __amdgpu_buffer_rsrc_t a_srd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(Ag) + a_ref, 0, (int)a_left, 0x00020000);
  const long long w_ref = ((long long)phase * d.Wrows + (long long)tn * BN) * d.Kpad;   // PH4: [parity][Wrows][Kpad]
  long long w_left = ((long long)(phase + 1) * d.Wrows * d.Kpad - w_ref) * 2;
  if (w_left > 0x7ffffe00LL) w_left = 0x7ffffe00LL;
  if (w_left < 0) w_left = 0;
  const __amdgpu_buffer_rsrc_t w_srd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(Wg) + w_ref, 0, (int)w_left, 0x00020000);

  int a_voff[LA];
  unsigned a_mask[LA];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int m = tm * BM + srow + 128 * i;
    const bool ok = m < M;
    if (conv) {
      const int hw = rHW;
      const int mm = ok ? m : 0;
      const int b = mm / hw;
      const int rem = mm - b * hw;
      const int oy = rem / rW, ox = rem - oy * rW;
      const int iy0 = oy * d.stride - py, ix0 = ox * d.stride - px;
      unsigned mk = 0;
      if constexpr (PH4) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int iy = iy0 + (t >> 1), ix = ix0 + (t & 1);
          if (ok && iy >= 0 && iy < p.Hv && ix >= 0 && ix < p.Wv) mk |= 1u << t;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = iy0 + t / 3, ix = ix0 + t % 3;
          if (ok && iy >= 0 && iy < p.Hv && ix >= 0 && ix < p.Wv) mk |= 1u << t;
        }
      }
      int sy = iy0, sx = ix0;
      if (UPS) {
        mk |= (unsigned)(ix0 & 1) << 16 | (unsigned)(iy0 & 1) << 17;
        sy >>= 1;
        sx >>= 1;
      }
      a_mask[i] = mk;
      const long long e = ((long long)b * d.Hi * d.Wi + (long long)sy * d.Wi + sx) * d.Cin + cch - a_ref;
      a_voff[i] = (int)(e * 2);
    } else {
      a_mask[i] = ok ? 1u : 0u;
      a_voff[i] = (int)(((long long)(ok ? m : 0) * d.lda + cch - a_ref) * 2);
    }
  }
  int w_voff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int nl = srow + 128 * i;
    const bool ok = tn * BN + nl < d.Wrows && nl < BN;
    w_voff[i] = ok ? (int)(((long long)nl * d.Kpad + cch) * 2) : OOB;
  }

  // stage range of this block and the two staging cursors (activation rows run one issue ahead of the weight rows)
  const int nst_all = p.nkc * p.ntaps;
  const int st0 = ksp * p.st_per;
  const int nst = (nst_all - st0 < p.st_per) ? nst_all - st0 : p.st_per;
  int a_cc = st0 / p.ntaps, a_tap = st0 - a_cc * p.ntaps;  // tap-inner: the 9 taps of a 32-channel slice are consecutive
  int b_cc = a_cc, b_tap = a_tap;

#define ISSUE_A(SLOT_)                                                                                   \
  do {                                                                                                   \
    char* ab_ = smem + (SLOT_) * SLOT + wave * 1024;                                                     \
    const int ky_ = PH4 ? (a_tap >> 1) : ((a_tap * 11) >> 5), kx_ = PH4 ? (a_tap & 1) : a_tap - 3 * ky_; \
    const int koff_ = a_cc * ROWB;                                                                       \
    const unsigned tbit_ = 1u << a_tap;                                                                  \
    if constexpr (!UPS) {                                                                                \
      const int toff_ = (ky_ * d.Wi + kx_) * d.Cin * 2;                                                  \
      _Pragma("unroll") for (int i_ = 0; i_ < LA; ++i_) {                                                \
        const int v_ = (a_mask[i_] & tbit_) ? a_voff[i_] + toff_ : OOB;                                  \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_srd, (lptr_t)(ab_ + i_ * RND), 16, v_, koff_, 0, 0);  \
      }                                                                                                  \
    } else {                                                                                             \
      _Pragma("unroll") for (int i_ = 0; i_ < LA; ++i_) {                                                \
        const int dy_ = (ky_ + (int)((a_mask[i_] >> 17) & 1)) >> 1;                                      \
        const int dx_ = (kx_ + (int)((a_mask[i_] >> 16) & 1)) >> 1;                                      \
        const int v_ = (a_mask[i_] & tbit_) ? a_voff[i_] + (dy_ * d.Wi + dx_) * d.Cin * 2 : OOB;         \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_srd, (lptr_t)(ab_ + i_ * RND), 16, v_, koff_, 0, 0);  \
      }                                                                                                  \
    }                                                                                                    \
    if (++a_tap == p.ntaps) {                                                                            \
      a_tap = 0;                                                                                         \
      ++a_cc;                                                                                            \
    }                                                                                                    \
  } while (0)

#define ISSUE_B(SLOT_)                                                                                   \
  do {                                                                                                   \
    char* bb_ = smem + (SLOT_) * SLOT + A_BYTES + wave * 1024;                                           \
    const int woff_ = (b_tap * p.nkc + b_cc) * ROWB;                                                     \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (lptr_t)(bb_), 16, w_voff[0], woff_, 0, 0);          \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (lptr_t)(bb_ + RND), 16, w_voff[1], woff_, 0, 0);    \
    if (grp == 0)                                                                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (lptr_t)(bb_ + 2 * RND), 16, w_voff[2], woff_, 0, 0); \
    if (++b_tap == p.ntaps) {                                                                            \
      b_tap = 0;                                                                                         \
      ++b_cc;                                                                                            \
    }                                                                                                    \
  } while (0)

// "stage s + 1 landed": everything issued after its last copy may stay in flight — C2: stage s + 2 is issued completely
// (LA + LB copies), C3: the activation rows of stage s + 3 too (LA).  LB = 3 (group 0) / 2 (group 1).
#define WAIT_NEXT(C2, C3)                                                     \
  do {                                                                        \
    if (C3) {                                                                 \
      if (grp == 0) wait_vm8<7>(); else wait_vm8<6>();                        \
    } else if (C2) {                                                          \
      if (grp == 0) wait_vm8<5>(); else wait_vm8<4>();                        \
    } else {                                                                  \
      wait_vm8<0>();                                                          \
    }                                                                         \
  } while (0)

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: A0 B0 A1 B1 A2 (the order the steady state would have issued them in) ----
  ISSUE_A(0);
  ISSUE_B(0);
  if (nst > 1) {
    ISSUE_A(1);
    ISSUE_B(1);
  }
  if (nst > 2) ISSUE_A(2);
  WAIT_NEXT(nst > 1, nst > 2);   // stage 0 landed (this wave's share)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  if (!ONEBAR && grp == 1) asm volatile("s_barrier" ::: "memory");  // group 1 runs one barrier behind group 0

  const int a_frag = (wm * 32 * MI + lq) * ROWB;
  const int b_frag = A_BYTES + (wn * 32 * NJ + lq) * ROWB;
  const int sw = (lq >> 2) & 3;
  const int co0 = ((0 + hi) ^ sw) * 16, co1 = ((2 + hi) ^ sw) * 16;
  constexpr int FSTR = 32 * ROWB;

  typename T::vec8 xf[MI], wf[NJ];

// one phase: Q = ring slot (literal), KS = MFMA k-step of the stage (literal), S = stage index (runtime, relative)
#define PHASE(Q, KS, S, FULL)                                                                               \
  do {                                                                                                      \
    if (!NOREAD || (S) == 0) {                                                                              \
      const char* base_ = smem + (Q) * SLOT + ((KS) ? co1 : co0);                                           \
      _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_) wf[j_] =                                            \
          *reinterpret_cast<const typename T::vec8*>(base_ + b_frag + j_ * FSTR);                           \
      _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) xf[i_] =                                            \
          *reinterpret_cast<const typename T::vec8*>(base_ + a_frag + i_ * FSTR);                           \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (!NOSTAGE) {                                                                                         \
      if ((KS) == 0) {                                                                                      \
        if ((FULL) || (S) + 2 < nst) ISSUE_B(((Q) + 2) & 3);                                                \
      } else {                                                                                              \
        if ((FULL) || (S) + 3 < nst) ISSUE_A(((Q) + 3) & 3);                                                \
        if (FULL) WAIT_NEXT(true, true);                                                                    \
        else if ((S) + 1 < nst) WAIT_NEXT((S) + 2 < nst, (S) + 3 < nst);                                    \
      }                                                                                                     \
    }                                                                                                       \
    if (LGK_EARLY) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
    TS(t_r);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    asm volatile("s_barrier" ::: "memory");                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    TS(t_b1);                                                                                               \
    if (!NOPRIO) __builtin_amdgcn_s_setprio(1);                                                             \
    if (NOMFMA) {                                                                                           \
      _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_) asm volatile("" ::"v"(wf[j_]));                     \
      _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) asm volatile("" ::"v"(xf[i_]));                     \
    } else {                                                                                                \
      _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                                     \
        _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_)                                                   \
          acc[i_][j_] = T::mfma32(wf[j_], xf[i_], acc[i_][j_]); /* D[n][m] */                               \
    }                                                                                                       \
    if (!NOPRIO) __builtin_amdgcn_s_setprio(0);                                                             \
    TS(t_m);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    asm volatile("s_barrier" ::: "memory");                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    TS(t_b2);                                                                                               \
  } while (0)
#define STAGE2(Q, S, FULL) \
  do {                     \
    PHASE(Q, 0, S, FULL);  \
    PHASE(Q, 1, S, FULL);  \
  } while (0)

  unsigned long long t_r = 0, t_b1 = 0, t_m = 0, t_b2 = 0, t_prev = 0;
#define TS(ACC)                                                     \
  do {                                                              \
    if (TIMING) {                                                   \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      ACC += now_ - t_prev;                                         \
      t_prev = now_;                                                \
    }                                                               \
  } while (0)
  if (TIMING) t_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long t_start = t_prev;
  const unsigned long long rt_start = TIMING ? __builtin_amdgcn_s_memrealtime() : 0;  // 100 MHz reference clock
  int st = 0;
  if constexpr (!ONEBAR) {
  // steady state: every stage of the iteration still has three successors -> no predicates, literal wait counts
  for (; st + 7 <= nst; st += 4) {
    STAGE2(0, st, 1);
    STAGE2(1, st + 1, 1);
    STAGE2(2, st + 2, 1);
    STAGE2(3, st + 3, 1);
  }
  // the last <= 6 stages (st is a multiple of 4: slots stay literal)
  if (st < nst) STAGE2(0, st, 0);
  if (st + 1 < nst) STAGE2(1, st + 1, 0);
  if (st + 2 < nst) STAGE2(2, st + 2, 0);
  if (st + 3 < nst) STAGE2(3, st + 3, 0);
  if (st + 4 < nst) STAGE2(0, st + 4, 0);
  if (st + 5 < nst) STAGE2(1, st + 5, 0);
  if (grp == 0) asm volatile("s_barrier" ::: "memory");  // pairs group 1's extra barrier
  } else {
  // ---- ONEBAR: ONE workgroup barrier per phase.  Between barriers k and k + 1 group 0 runs [M(k) ; R(k + 1)] and group 1
  // runs [R(k) ; M(k)]: the first half of the interval is M0 || R1, the second R0 || M1 — the same complementary pairing as
  // above with half the barriers (the pairing inside an interval is by equal section lengths, not enforced).
  //   RAW: "stage s landed" is waited for in front of barrier 2s - 1 by every wave (group 0 in its R(s - 1, 1), which runs
  //        in interval 2s - 2; group 1 behind its M(s - 1, 0), same interval); the first read of stage s is group 0's R(s, 0)
  //        in interval 2s - 1.
  //   WAR: B(s + 2) -> slot of stage s - 2, last read by group 1 in interval 2s - 3, issued in interval 2s - 1 (group 0) /
  //        2s (group 1); A(s + 3) -> slot of stage s - 1, last read by group 1 in interval 2s - 1 (retired before its MFMAs),
  //        issued in interval 2s (group 0) / 2s + 1 (group 1).
#define RPART(Q, KS, S, FULL, G)                                                                            \
  do {                                                                                                      \
    {                                                                                                       \
      const char* base_ = smem + (Q) * SLOT + ((KS) ? co1 : co0);                                           \
      _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_) wf[j_] =                                            \
          *reinterpret_cast<const typename T::vec8*>(base_ + b_frag + j_ * FSTR);                           \
      _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) xf[i_] =                                            \
          *reinterpret_cast<const typename T::vec8*>(base_ + a_frag + i_ * FSTR);                           \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if ((KS) == 0) {                                                                                        \
      if ((FULL) || (S) + 2 < nst) ISSUE_B(((Q) + 2) & 3);                                                  \
    } else {                                                                                                \
      if ((FULL) || (S) + 3 < nst) ISSUE_A(((Q) + 3) & 3);                                                  \
      if ((G) == 0) {                                                                                       \
        if (FULL) wait_vm8<7>();                                                                            \
        else if ((S) + 1 < nst) { if ((S) + 3 < nst) wait_vm8<7>(); else if ((S) + 2 < nst) wait_vm8<5>(); else wait_vm8<0>(); } \
      }                                                                                                     \
    }                                                                                                       \
    TS(t_r);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  } while (0)
#define MPART()                                                                                             \
  do {                                                                                                      \
    if (!NOPRIO) __builtin_amdgcn_s_setprio(1);                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                                       \
      _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_)                                                     \
        acc[i_][j_] = T::mfma32(wf[j_], xf[i_], acc[i_][j_]); /* D[n][m] */                                 \
    if (!NOPRIO) __builtin_amdgcn_s_setprio(0);                                                             \
    TS(t_m);                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  } while (0)
#define BAR1()                                   \
  do {                                           \
    __builtin_amdgcn_sched_barrier(0);           \
    asm volatile("s_barrier" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);           \
    TS(t_b1);                                    \
  } while (0)
// group 1, behind M(s, 0): its share of stage s + 1 landed (only B(s + 2), issued in this phase, may stay in flight: LB = 2
// for this group, plus A(s + 2) issued one phase earlier: LA + LB = 4)
#define WAIT_G1(S, FULL)                                                      \
  do {                                                                        \
    if (FULL) wait_vm8<4>();                                                  \
    else if ((S) + 1 < nst) { if ((S) + 2 < nst) wait_vm8<4>(); else wait_vm8<0>(); } \
  } while (0)
    if (grp == 0) {
      RPART(0, 0, 0, 0, 0);
#define G0_STAGE(Q, S, FULL)                                         \
  do {                                                               \
    BAR1();                                                          \
    MPART();                                                         \
    RPART(Q, 1, S, FULL, 0);                                         \
    BAR1();                                                          \
    MPART();                                                         \
    if ((FULL) || (S) + 1 < nst) RPART(((Q) + 1) & 3, 0, (S) + 1, FULL, 0); \
  } while (0)
      for (; st + 7 <= nst; st += 4) {
        G0_STAGE(0, st, 1);
        G0_STAGE(1, st + 1, 1);
        G0_STAGE(2, st + 2, 1);
        G0_STAGE(3, st + 3, 1);
      }
      if (st < nst) G0_STAGE(0, st, 0);
      if (st + 1 < nst) G0_STAGE(1, st + 1, 0);
      if (st + 2 < nst) G0_STAGE(2, st + 2, 0);
      if (st + 3 < nst) G0_STAGE(3, st + 3, 0);
      if (st + 4 < nst) G0_STAGE(0, st + 4, 0);
      if (st + 5 < nst) G0_STAGE(1, st + 5, 0);
#undef G0_STAGE
    } else {
#define G1_STAGE(Q, S, FULL)        \
  do {                              \
    BAR1();                         \
    RPART(Q, 0, S, FULL, 1);        \
    MPART();                        \
    WAIT_G1(S, FULL);               \
    BAR1();                         \
    RPART(Q, 1, S, FULL, 1);        \
    MPART();                        \
  } while (0)
      for (; st + 7 <= nst; st += 4) {
        G1_STAGE(0, st, 1);
        G1_STAGE(1, st + 1, 1);
        G1_STAGE(2, st + 2, 1);
        G1_STAGE(3, st + 3, 1);
      }
      if (st < nst) G1_STAGE(0, st, 0);
      if (st + 1 < nst) G1_STAGE(1, st + 1, 0);
      if (st + 2 < nst) G1_STAGE(2, st + 2, 0);
      if (st + 3 < nst) G1_STAGE(3, st + 3, 0);
      if (st + 4 < nst) G1_STAGE(0, st + 4, 0);
      if (st + 5 < nst) G1_STAGE(1, st + 5, 0);
#undef G1_STAGE
    }
#undef RPART
#undef MPART
#undef BAR1
#undef WAIT_G1
  }
  if (TIMING && p.ws && lane == 0 && p.splitk <= 1) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws) + ((long long)blockIdx.x * 8 + wave) * 8;
    o[0] = t_r; o[1] = t_b1; o[2] = t_m; o[3] = t_b2;
    o[4] = __builtin_amdgcn_s_memtime() - t_start; o[5] = nst; o[6] = grp;
    o[7] = __builtin_amdgcn_s_memrealtime() - rt_start;
    o[64 * 2048 + 0] = rt_start - rt_kernel;   // prologue (x 10 ns)
  }
#undef TS
#undef STAGE2
#undef PHASE
#undef WAIT_NEXT
#undef ISSUE_A
#undef ISSUE_B

  // ======================================================================================================
  // split-K inside the launch: publish the accumulators, ticket, last arriver reduces in slice order
  // ======================================================================================================
  if (p.splitk > 1) {
    constexpr int SLAB_F4 = BM * BN / 4;  // float4 per slab (40 per thread)
    float* slab0 = p.ws + (long long)tile * p.splitk * (SLAB_F4 * 4);
    const __amdgpu_buffer_rsrc_t s_srd =
        __builtin_amdgcn_make_buffer_rsrc(slab0, 0, p.splitk * SLAB_F4 * 16, 0x00020000);
    {
      const int soff = ksp * (SLAB_F4 * 16);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int q = (i * NJ + j) * 4 + g;
            const f32x4 v = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v),
                                                   s_srd, (q * 512 + tid) * 16, soff, /*sc1: write-through*/ 16);
          }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(p.cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = (int)t;
    }
    __syncthreads();
    const int ticket = *flag;
    if (ticket != p.splitk - 1) return;
    // last arriver: S == 2 -> own + other (commutative: independent of who is last); S > 2 -> all slabs in slice
    // order (the own one was published like the others), so the f32 sum does not depend on the arrival order
    if (p.splitk > 2) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    for (int sl = 0; sl < p.splitk; ++sl) {
      if (p.splitk == 2 && sl == ksp) continue;
      const int soff = sl * (SLAB_F4 * 16);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int q = (i * NJ + j) * 4 + g;
            const f32x4 v = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(s_srd, (q * 512 + tid) * 16, soff, /*sc1*/ 16));
            acc[i][j][4 * g + 0] += v[0];
            acc[i][j][4 * g + 1] += v[1];
            acc[i][j][4 * g + 2] += v[2];
            acc[i][j][4 * g + 3] += v[3];
          }
    }
  }

  // ======================================================================================================
  // epilogue: f32 math in registers -> 16-bit half tile (128 rows) in LDS -> 16-byte row stores (+ residual, + stats)
  // ======================================================================================================
  const int N = d.N;
  constexpr int CS_LD = BN + 8;          // halfs per staged row (656 B: a multiple of 16)
  constexpr int CPR = BN / 8, RL = 12;   // 16-byte chunks per row, row lanes (480 of the 512 threads stream rows out)
  constexpr int NR = (128 + RL - 1) / RL;  // rows per thread and pass (11; the last one only for rl < 128 - 10 * RL)
  u16* Cs = reinterpret_cast<u16*>(smem);
  const u16* __restrict__ RV = reinterpret_cast<const u16*>(d.rowvec);
  const int ch = tid % CPR, rl = tid / CPR;
  const int ncol = tn * BN + ch * 8;
  const u16* __restrict__ Rg = d.R ? reinterpret_cast<const u16*>(d.R) : nullptr;
  u16* __restrict__ Cg = reinterpret_cast<u16*>(d.C);
  ColStat cstat;
  colstat_init(cstat);
  const bool want_stats = d.stats != nullptr;
  const bool streamer = rl < RL && ncol < N;
  const bool full_chunk = ncol + 8 <= N;
  // output / residual rows through tile-local buffer descriptors: 32-bit row offsets, rows >= M clipped by the hardware
  // (loads return zero, stores are dropped)
  const long long rows_left = (long long)M - (long long)tm * BM;
  auto clip31 = [](long long b) { return (int)(b > 0x7ffffe00LL ? 0x7ffffe00LL : (b < 0 ? 0 : b)); };
  // PH4: a tile's rows are low-resolution pixels scattered over the whole output -> one descriptor over the output tensor
  // (dbir_gemm_8p_eligible bounds it below 2 GB), per-row offsets from orow_off()
  const __amdgpu_buffer_rsrc_t c_srd = __builtin_amdgcn_make_buffer_rsrc(
      PH4 ? Cg : Cg + (long long)tm * BM * d.ldc, 0,
      clip31(((PH4 ? (long long)d.M - 1 : rows_left - 1) * d.ldc + N) * 2), 0x00020000);
  // PH4: element offset (from Cg) of the output row that tile row `trow` (low-resolution pixel m = tm * BM + trow) produces
  // for this tile's parity; Hi / Wi are powers of two (eligibility).  -1 for rows past the end.
  auto orow_off = [&](int trow) -> long long {
    const int m = tm * BM + trow;
    if (m >= M) return -1;
    const int j = m & (d.Wi - 1), i = (m >> p.lgW) & (d.Hi - 1), b = m >> p.lgHW;
    return ((long long)(b * d.Ho + 2 * i + pa) * d.Wo + 2 * j + pb) * d.ldc;
  };
  const __amdgpu_buffer_rsrc_t r_srd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(Rg ? Rg + (long long)tm * BM * d.ldr : Cg), 0, Rg ? clip31(((rows_left - 1) * d.ldr + N) * 2) : 0,
      0x00020000);
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

  // Column offsets of the f32 math: bias[n] + rowvec[sample][n] per sample the tile touches, built ONCE in LDS (behind the
  // staged half tile) — the register phase below then reads 16 bytes from LDS per (j, g) instead of waiting for two
  // dependent global loads each (that chain, not bandwidth, was most of the epilogue: 33 us of a 135 us launch).
  constexpr int TAB_OFF = 128 * CS_LD * 2;                        // bytes: behind the staged 128 x 328 half tile
  float* tab = reinterpret_cast<float*>(smem + TAB_OFF);          // [samples][BN]
  const int rpb = RV ? d.rows_per_batch : 0x7fffffff;
  const int m_first = tm * BM, m_last = (tm * BM + BM - 1 < M ? tm * BM + BM - 1 : M - 1);
  const int samp0 = m_first / rpb, nsamp = m_last / rpb - samp0 + 1;
  auto tab_value = [&](int q) {
    const int sidx = q / BN, col = q - sidx * BN, n = tn * BN + col;
    float v = 0.f;
    if (n < N) {
      if (d.bias) v = d.bias[n];
      if (RV) v += T::to_f32(RV[(long long)(samp0 + sidx) * d.rowvec_ld + n]);
    }
    return v;
  };
  constexpr int TQ = 3;   // <= 4 samples per tile (the usual case): this thread's entries are requested in front of the barrier
  float tv[TQ];
  const bool tab_regs = nsamp * BN <= TQ * 512;
  if (tab_regs) {
#pragma unroll
    for (int k = 0; k < TQ; ++k) tv[k] = tid + 512 * k < nsamp * BN ? tab_value(tid + 512 * k) : 0.f;
  }
  __syncthreads();  // operand ring dead
  if (tab_regs) {
#pragma unroll
    for (int k = 0; k < TQ; ++k)
      if (tid + 512 * k < nsamp * BN) tab[tid + 512 * k] = tv[k];
  } else {
    for (int q = tid; q < nsamp * BN; q += 512) tab[q] = tab_value(q);
  }

  // Pass h stages accumulator block i = h of EVERY wave: staged row lr = wm * 32 + lq <-> tile row wm * 64 + h * 32 + lq.
  auto pass = [&](auto hc) {
    constexpr int h = decltype(hc)::value;
    // residual rows of this pass: all loads in flight before the LDS round trip (they were the epilogue's critical path:
    // one dependent HBM / L2 round trip per row)
    u32x4 rres[NR];
    if (Rg && streamer && full_chunk) {
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int lr = rl + RL * k;
        const int trow = (lr >> 5) * 64 + h * 32 + (lr & 31);
        rres[k] = __builtin_amdgcn_raw_buffer_load_b128(r_srd, lr < 128 ? (trow * (int)d.ldr + ncol) * 2 : OOB, 0, 0);
      }
    }
    __syncthreads();  // table built (h = 0) / previous half streamed out (h = 1)
    {
      const int lrow = wm * 32 + lq;
      const int m = tm * BM + wm * 64 + h * 32 + lq;
      const int mb = (m < M ? m : M - 1);
      const float* trow_ = tab + (mb / rpb - samp0) * BN + wn * 32 * NJ + 4 * hi;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn * 32 * NJ + j * 32 + 8 * g + 4 * hi;
          const float4 bb = *reinterpret_cast<const float4*>(trow_ + j * 32 + 8 * g);
          float v[4] = {acc[h][j][4 * g + 0] + bb.x, acc[h][j][4 * g + 1] + bb.y, acc[h][j][4 * g + 2] + bb.z,
                        acc[h][j][4 * g + 3] + bb.w};
          if (d.act == DBIR_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
          } else if (d.act == DBIR_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
          } else if (d.act == DBIR_ACT_LRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * d.act_param;
          }
          uint2 pk;
          pk.x = (uint32_t)T::from_f32(v[0] * d.out_scale) | ((uint32_t)T::from_f32(v[1] * d.out_scale) << 16);
          pk.y = (uint32_t)T::from_f32(v[2] * d.out_scale) | ((uint32_t)T::from_f32(v[3] * d.out_scale) << 16);
          *reinterpret_cast<uint2*>(Cs + lrow * CS_LD + nl) = pk;
        }
      }
    }
    __syncthreads();
    if (streamer) {
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int lr = rl + RL * k;
        const int trow = (lr >> 5) * 64 + h * 32 + (lr & 31);
        const int m = tm * BM + trow;
        if (lr >= 128) continue;
        float a[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(Cs + lr * CS_LD + ch * 8), a);
        long long oro = 0;   // PH4: output row offset (elements)
        if constexpr (PH4) oro = orow_off(trow);
        if (full_chunk) {
          if (Rg) {
            float b[8];
            unpack8<T>(__builtin_bit_cast(uint4, rres[k]), b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];
          }
          const uint4 v = pack8<T>(a);
          const int so = PH4 ? (oro < 0 ? OOB : (int)((oro + ncol) * 2)) : (trow * (int)d.ldc + ncol) * 2;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), c_srd, so, 0, 0);
          if (want_stats && m < M) {
            unpack8<T>(v, a);  // statistics of what was stored (16-bit rounded)
            colstat_add(cstat, a);
          }
        } else if (m < M) {
          const long long ro = PH4 ? oro : (long long)m * d.ldc;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = a[e];
            if (ncol + e < N) {
              if (Rg) x += T::to_f32(Rg[(long long)m * d.ldr + ncol + e]);
              const u16 hv = T::from_f32(x);
              Cg[ro + ncol + e] = hv;
              x = T::to_f32(hv);
            }
            a[e] = x;
          }
          if (want_stats) colstat_add(cstat, a);
        }
      }
    }
  };
  const unsigned long long rt_e0 = TIMING ? __builtin_amdgcn_s_memrealtime() : 0;
  pass(std::integral_constant<int, 0>{});
  const unsigned long long rt_e1 = TIMING ? __builtin_amdgcn_s_memrealtime() : 0;
  pass(std::integral_constant<int, 1>{});
  if (TIMING && p.ws && lane == 0 && p.splitk <= 1) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws) + ((long long)blockIdx.x * 8 + wave) * 8 + 64 * 2048;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long rt_e2 = __builtin_amdgcn_s_memrealtime();
    o[1] = rt_e2 - rt_kernel;  // whole kernel (x 10 ns)
    o[2] = rt_e0 - rt_kernel;  // until the table is built
    o[3] = rt_e1 - rt_e0;      // pass 0 (its stores issued)
    o[4] = rt_e2 - rt_e1;      // pass 1 + store drain
  }
  if (want_stats) {
    // PH4: the 4 * st_tps tiles of a sample (4 parities x st_tps low-resolution row tiles) are adjacent statistics rows
    const long long srow = PH4 ? (long long)(tm / p.st_tps) * (4 * p.st_tps) + phase * p.st_tps + tm % p.st_tps : tm;
    colstat_finish<512, BN, CPR, RL>(cstat, streamer, smem, tid, ch, rl, d.stats + srow * 2 * N, tn, N);
  }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

// Can the fine-phase kernel (tiles 80 ..) run this descriptor?  (validated by dbir_gemm already)
bool dbir_gemm_8p_eligible(const dbir_gemm_desc& d) {
  if (d.out_f32 || d.store_mode != 0 || d.act == DBIR_ACT_GEGLU || d.batch > 1) return false;
  if (d.rowvec && d.rows_per_batch < 8) return false;  // (bias + row vector) table: <= 33 samples per 256-row tile
  if (d.ldc % 8 != 0 || (reinterpret_cast<uintptr_t>(d.C) & 15)) return false;
  if (d.R && (d.ldr % 8 != 0 || (reinterpret_cast<uintptr_t>(d.R) & 15))) return false;
  if ((reinterpret_cast<uintptr_t>(d.A) & 15) || (reinterpret_cast<uintptr_t>(d.W) & 15)) return false;
  if (d.mode == DBIR_MODE_LINEAR) {
    if (d.K % 32 != 0 || d.lda % 8 != 0) return false;
  } else {
    if (d.Cin % 32 != 0 || d.lda != d.Cin) return false;
    if ((long long)d.B * d.Hi * d.Wi >= 2147483647LL) return false;
    if (d.upsample == 2) {   // PH4: four 2x2 convolutions on the low-resolution input, rows scattered to the output
      auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
      if (d.R || d.rowvec || d.stride != 1 || d.pad != 1 || d.Ho != 2 * d.Hi || d.Wo != 2 * d.Wi) return false;
      if (!pow2(d.Hi) || !pow2(d.Wi) || d.K != 4 * d.Cin) return false;
      if (((long long)d.M - 1) * d.ldc + d.N >= (1LL << 30)) return false;           // 32-bit byte offsets into the output
      if (4LL * d.Wrows * d.Kpad * 2 >= 0x7ffffe00LL) return false;
    }
  }
  return true;
}

// bytes of workspace a split-K launch of this kernel needs (slabs + arrival counters)
long long dbir_gemm_8p_ws_bytes(int M, int N, int splitk) {   // (PH4: pass M = 4 * ceil(B Hi Wi / 256) * 256)
  const long long tiles = (long long)cdiv(M, kBM) * cdiv(N, kBN);
  return tiles * splitk * (long long)kBM * kBN * 4 + tiles * 4 + 256;
}

template <typename T, int VAR>
static int launch_8p(P8& p, hipStream_t s) {
  constexpr int lds = 4 * (kBM + kBN) * 64;  // 147456: the ring; the epilogue's half tile (128 x 328 x 2 = 83968) reuses it
  const dbir_gemm_desc& dd = p.d;
  constexpr bool PH4 = (VAR & 256) != 0;
  p.ntaps = dd.mode == DBIR_MODE_LINEAR ? 1 : (PH4 ? 4 : 9);
  p.nkc = (dd.mode == DBIR_MODE_LINEAR ? dd.K : dd.Cin) / 32;
  p.mtiles = cdiv(PH4 ? dd.B * dd.Hi * dd.Wi : dd.M, kBM);
  p.ntiles = cdiv(dd.N, kBN);
  p.ph_tiles = PH4 ? p.mtiles * p.ntiles : 0;
  p.st_tps = 1;
  if (PH4) {
    p.lgW = __builtin_ctz((unsigned)dd.Wi);
    p.lgHW = p.lgW + __builtin_ctz((unsigned)dd.Hi);
    p.Hv = dd.Hi;
    p.Wv = dd.Wi;
  }
  const int tiles = p.mtiles * p.ntiles * (PH4 ? 4 : 1);
  const int nst_all = p.nkc * p.ntaps;
  int sk = dd.splitk > 1 ? dd.splitk : 1;
  if (sk > nst_all) sk = nst_all;
  p.st_per = cdiv(nst_all, sk);
  p.splitk = cdiv(nst_all, p.st_per);
  p.ws = nullptr;
  p.cnt = nullptr;
  if (p.splitk > 1) {
    const long long slab_bytes = (long long)tiles * p.splitk * kBM * kBN * 4;
    const long long need = slab_bytes + (long long)tiles * 4;
    if (!dd.ws || dd.ws_bytes < need || (reinterpret_cast<uintptr_t>(dd.ws) & 15)) {
      dbir_set_error("dbir_gemm: tile 80 split-K %d needs a 16-byte aligned workspace of %lld bytes (got %lld)", p.splitk,
                     need, dd.ws_bytes);
      return DBIR_ERR_ARG;
    }
    if ((long long)p.splitk * kBM * kBN * 4 >= 0x7ffffe00LL) {
      dbir_set_error("dbir_gemm: tile 80 split-K %d too large", p.splitk);
      return DBIR_ERR_ARG;
    }
    p.ws = reinterpret_cast<float*>(dd.ws);
    p.cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(dd.ws) + slab_bytes);
    if (hipMemsetAsync(p.cnt, 0, (size_t)tiles * 4, s) != hipSuccess) {
      dbir_set_error("dbir_gemm: tile 80: hipMemsetAsync of the split-K counters failed");
      return DBIR_ERR_LAUNCH;
    }
  }
  if ((VAR & 32) && p.splitk <= 1) p.ws = reinterpret_cast<float*>(dd.ws);  // diagnostic timing dump
  if (p.d.stats) {
    if (PH4) {   // whole 256-row tiles per sample and parity
      if ((dd.Hi * dd.Wi) % kBM == 0) {
        p.st_tps = dd.Hi * dd.Wi / kBM;
        g_dbir_stats_rows = kBM;
      } else {
        p.d.stats = nullptr;
      }
    } else if (dd.M % kBM == 0) {
      g_dbir_stats_rows = kBM;
    } else {
      p.d.stats = nullptr;
    }
  }
  auto kern = &gemm8p_kernel<T, VAR>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * p.splitk)), dim3(512), lds, s, p);
  DBIR_CHECK_LAUNCH("dbir_gemm(8p)");
  return DBIR_OK;
}

int dbir_gemm_8p(const dbir_gemm_desc& dd, int Hv, int Wv, int tile, hipStream_t s) {
  P8 p;
  memset(&p, 0, sizeof(p));
  p.d = dd;
  p.Hv = Hv;
  p.Wv = Wv;
  p.vec_bias = dd.bias && (reinterpret_cast<uintptr_t>(dd.bias) & 15) == 0;
  p.vec_rv = dd.rowvec && (reinterpret_cast<uintptr_t>(dd.rowvec) & 7) == 0 && dd.rowvec_ld % 4 == 0;
  p.a_elems = dd.mode == DBIR_MODE_CONV3X3 ? (long long)dd.B * dd.Hi * dd.Wi * dd.Cin
                                           : (long long)(dd.M - 1) * dd.lda + dd.K;
  if (tile != 80) {
    dbir_set_error("dbir_gemm: bad fine-phase tile %d", tile);
    return DBIR_ERR_ARG;
  }
  if (dd.mode == DBIR_MODE_CONV3X3 && dd.upsample == 2)
    return dd.dtype == DBIR_F16 ? launch_8p<F16, 256>(p, s) : launch_8p<BF16, 256>(p, s);
  if (dd.mode == DBIR_MODE_CONV3X3 && dd.upsample)
    return dd.dtype == DBIR_F16 ? launch_8p<F16, 1>(p, s) : launch_8p<BF16, 1>(p, s);
#ifdef P8_VARIANTS
  {
    static const int var = getenv("DBIR_P8_VAR") ? atoi(getenv("DBIR_P8_VAR")) : 0;
    if (var && dd.dtype == DBIR_F16) {
      switch (var) {
        case 2: return launch_8p<F16, 2>(p, s);
        case 4: return launch_8p<F16, 4>(p, s);
        case 8: return launch_8p<F16, 8>(p, s);
        case 16: return launch_8p<F16, 16>(p, s);
        case 32: return launch_8p<F16, 32>(p, s);
        case 34: return launch_8p<F16, 34>(p, s);
        case 64: return launch_8p<F16, 64>(p, s);
        case 66: return launch_8p<F16, 66>(p, s);
        case 12: return launch_8p<F16, 12>(p, s);
        case 128: return launch_8p<F16, 128>(p, s);
        case 130: return launch_8p<F16, 130>(p, s);
        case 160: return launch_8p<F16, 160>(p, s);
        case 20: return launch_8p<F16, 20>(p, s);
      }
    }
  }
#endif
  return dd.dtype == DBIR_F16 ? launch_8p<F16, 0>(p, s) : launch_8p<BF16, 0>(p, s);
}
