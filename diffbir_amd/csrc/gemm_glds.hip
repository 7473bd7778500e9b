// Implicit-GEMM (linear / 1x1 / 3x3 convolution) on gfx950 MFMA, direct-to-LDS staged, multi-slot ring.
//
// This is the hot kernel family of the engine: every conv3x3 with Cin % 64 == 0 and every linear with K % 64 == 0
// of the UNet / ControlNet / VAE / SwinIR (i.e. all of them except the 4-/8-channel stem convs) runs here.
// The generic register-staged kernel in gemm.hip remains as the fallback for the odd shapes and f32 stores.
//
// Structure (cdna_hip_programming.md §5 / T2 / T3+T4, re-derived for this engine's operand layout):
//   * WM x WN wave64 per block, each wave owns a (32*MI) x (32*NJ) output tile of v_mfma_f32_32x32x16
//     accumulators; K depth of a tile BKT = 64 halfs (one 128-byte line per tile row) or 32.  One template gives the
//     128x128 / 256x64 / 64x256 / 256x128 / 256x256 and the 160-wide (256x160, 128x160: the UNet's channel counts are
//     multiples of 320) tiles; see dispatch2() for the catalogue.
//   * Both operand tiles go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (16 B per lane, no VGPR round trip)
//     through two block-local buffer descriptors: per-lane 32-bit offsets are computed once per block, the
//     (tap, channel tile) part of a K tile is a scalar offset, and padding / out-of-tile rows use an out-of-range
//     offset for which the hardware returns zeros.
//     The LDS image is lane-linear, so the bank-conflict swizzle is applied on the SOURCE offset: LDS chunk
//     position `cpos` of row r holds logical 16-byte chunk `cpos ^ key(r)` (key = (r >> 1) & 7 for 128-byte rows,
//     (r >> 2) & 3 for 64-byte rows); the MFMA fragment ds_read_b128 applies the same XOR (guide rule 21), which
//     makes the 16 lanes of every ds_read_b128 service group hit 16 distinct 16-byte slots: conflict-free.
//   * conv3x3 is an implicit GEMM over K = (tap, channel); Cin % 64 == 0 makes every K tile lie inside one tap, so a
//     tile row is one contiguous run of the NHWC input at a per-row pixel offset.  K tiles are visited tap-inner
//     (the 9 taps of one channel slice consecutively: the shifted windows re-hit L2).
//   * STAGES-slot LDS ring with counted `s_waitcnt vmcnt(N)` (never a drain in steady state) + raw s_barrier;
//     PIPE = 1 software-pipelines the LDS fragment reads; DEPH = 1 runs the block's two wave groups half a K tile
//     apart so that one group's staging burst (paced by the vector memory pipe) overlaps the other group's MFMAs.
//   * MFMA orientation is D[n][m] (first operand = weight rows) so that a lane holds 4 consecutive output
//     channels of one pixel: the f32 epilogue (bias, time-embedding row vector, SiLU/GELU/LeakyReLU/GEGLU,
//     scale) runs in registers, the 16-bit tile is transposed through LDS and written with 16-byte row-contiguous
//     stores (+ residual add on that side), or transposed per batch (V^T for the attention kernel).
//   * Split-K: K slices computed by different workgroups into f32 slabs + a deterministic reduce/epilogue kernel.
//   * blockIdx -> tile mapping is XCD-aware (bijective remap, guide T1): each XCD's L2 sees a contiguous
//     range of tiles with the N tiles of one activation panel adjacent.
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"

extern thread_local int g_dbir_stats_rows;  // gemm.hip

namespace {

constexpr int BK = 64;
#ifdef DBIR_DIAG  // diagnostic build (sh build.sh -DDBIR_DIAG): env DBIR_GEMM_DEBUG = 1 skip staging, 2 skip MFMAs,
                  // 5 s_memtime interval accumulators (de-phased kernels) into the workspace pointer
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif

struct G2Params {
  dbir_gemm_desc d;
  int Hv, Wv;   // virtual (upsampled) input extent for conv bounds checks
  int nkc;      // K tiles per tap (conv) / total K tiles (linear)
  int ntaps;    // 9 (conv) / 1 (linear)
  int mtiles, ntiles;
  int vec_bias, vec_rv;  // bias / row vector may be read with 16 B / 8 B vector loads
  int splitk, kt_per;    // split-K: number of K slices (1 = off) and K tiles per slice
  float* ws;             // split-K: f32 partial sums [z][slice][M][N]
  long long a_elems;     // elements of the activation tensor addressable from d.A (+ batch z offset)
  int group_m;           // tile order: GROUP_M row tiles x all column tiles per group (1 = column tiles fastest)
  int tap_inner;         // conv K order: 1 = (channel tile, tap), 0 = (tap, channel tile)
  int debug;             // DIAGNOSTIC (env DBIR_GEMM_DEBUG): 1 = skip steady-state staging, 2 = skip MFMAs
};

typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PIPE = 1: the MFMA fragments of k-step ks+1 are read from LDS BEFORE the MFMAs of k-step ks are issued (two
// register sets), so the ds_read latency of all but a K tile's first k-step hides under the matrix pipe; with
// PIPE = 0 the compiler emits read -> s_waitcnt lgkmcnt(0) -> MFMA per k-step and the two waves of a SIMD, which run
// in lockstep, expose that latency together (rocprofv3: 35 % MFMA utilisation, 43 % of wave cycles in s_waitcnt).
// DEPH = 1 (needs STAGES = 3, PIPE = 1): the block's waves form two groups that run HALF A K TILE apart — while group 0
// issues its share of the next tile's direct-to-LDS loads (a burst paced by the vector memory pipe: ~1000 cycles per
// K tile when all 8 waves issue at once, measured with s_memtime, during which no MFMA can issue because a wave
// issues in order), group 1 multiplies, and vice versa; two workgroup barriers per K tile.  Interval I_2k: group 0
// stages tile k+1 / group 1 multiplies tile k-1; I_2k+1: group 0 multiplies tile k / group 1 stages tile k+1.
//   RAW: group 0 waits vmcnt(LOADS) after its staging (its share of tile k landed), group 1 waits vmcnt(0) after its
//        multiply (its share of tile k, issued one interval earlier), both before the barrier that precedes the first
//        read of tile k;  WAR: tile k+1 refills the slot of tile k-2, last read by group 1 two intervals earlier.
// BKT = K depth of a tile (64, or 32: LDS rows of 64 B, 4 chunks; lets a 256x256 tile keep a 3-slot ring in 96 KB for
// the de-phased schedule).
// XPF = 1 (needs PIPE = 1, lockstep): cross-tile fragment prefetch — the wait + barrier that acquires K tile kt + 1 sits
// in front of the LAST k-step's MFMAs of tile kt (every fragment of tile kt is in registers by then), followed at once by
// the first fragment reads of tile kt + 1 and the refill of tile kt's slot: the LDS latency and the barrier skew of a tile
// boundary hide under MI x NJ armed MFMAs instead of standing between two tiles, and the ring runs one tile deeper (the
// schedule measured on the fused transformer kernels, xformer.hip: 1300 -> 1100 cycles per 10-MFMA tile).
// LDW > 0 (producer / consumer split, tiles 90 - 92): the block has LDW extra LOADER waves that issue every direct-to-LDS
// copy (and do nothing else) while the WM x WN matrix waves only read fragments and issue MFMAs.  A wave that issues the
// copies stalls in VMEM issue while the copy engine drains (~34 B/clk per CU) and, issuing in order, cannot start its MFMAs
// behind them; with the stall on separate waves the copy of K tile kt + 2 runs UNDER the MFMAs of tile kt instead of in
// front of them (tools/probes/lds_stream_bench.hip: 0.49 -> 0.37 us per 20 KB tile with 10 MFMAs per wave = the pure MFMA time).
// One workgroup barrier per K tile: loaders wait for tile kt (counted vmcnt) in front of it and refill the slot of tile
// kt - 1 behind it; matrix waves finish their fragment reads of tile kt - 1 (lgkmcnt(0)) in front of it.  Loaders leave
// after the K loop (an ended wave no longer takes part in barriers), the matrix waves run the epilogue.
template <typename T, int WM, int WN, int MI, int NJ, int STAGES, int MINW, int PIPE, int DEPH, int BKT, int XPF = 0,
          int LDW = 0>
__global__ __launch_bounds__(64 * (WM * WN + LDW), MINW) void gemm_glds_kernel(const G2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the body uses gfx950-only builtins (buffer descriptors, LDS-DMA, MFMA)
  constexpr int NT = 64 * WM * WN;             // matrix-wave threads (the epilogue's thread count)
  constexpr int NTS = LDW ? 64 * LDW : NT;     // threads that stage operand tiles
  constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
  constexpr int BK = BKT;                      // shadows the file-scope default inside the kernel
  constexpr int ROWB = BK * 2, CPR = BK / 8;   // bytes / 16-byte chunks per LDS tile row
  constexpr int KS = BK / 16;                  // MFMA k-steps per tile
  constexpr int RPP = NTS / CPR;               // tile rows covered by one glds pass of the staging threads
  constexpr int PASS_BYTES = NTS * 16;
  constexpr int RA = BM / RPP, RB = (BN + RPP - 1) / RPP;  // glds per thread per stage (activation / weight tile)
  constexpr int LOADS = RA + RB;
  // BN = 160 (NJ = 5) is not a multiple of the pass height: the weight region is rounded up to whole passes and the
  // rows past BN are fed from the zero page
  constexpr bool EXACT_B = DEPH && (BN % RPP != 0);  // DEPH needs 3 slots: no room for the round-up rows
  constexpr int A_BYTES = BM * ROWB, B_BYTES = (EXACT_B ? BN : RB * RPP) * ROWB, BUF_BYTES = A_BYTES + B_BYTES;
  static_assert(BM % RPP == 0, "activation tile rows must be a multiple of the pass height");
  static_assert(STAGES >= 2 && (STAGES - 1) * LOADS < 64, "vmcnt is 6 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const dbir_gemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lq = lane & 31, hi = lane >> 5;
  const int bz = blockIdx.y;
  const bool loader = LDW > 0 && wave >= WM * WN;
  const int stid = LDW ? tid - NT : tid;       // index among the staging threads (negative: a matrix wave of an LDW block)
  const int swave = LDW ? wave - WM * WN : wave;

  // ---- XCD-aware tile mapping (bijective) ----
  int tm, tn, ksp;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r = nwg & 7;
    int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int tiles = p.mtiles * p.ntiles;
    ksp = lid / tiles;  // K slice (0 when split-K is off)
    lid -= ksp * tiles;
    // Tile order inside an XCD's contiguous range (round 4): the ~32 tiles an XCD runs AT THE SAME TIME stream K in step, so
    // what its L2 must hold is one K window of every DISTINCT operand panel among them.  With column tiles fastest a
    // wide problem (N / BN >= 8) runs 1 activation panel against 32 weight panels; groups of GROUP_M row tiles x all
    // column tiles, row tile fastest, make the concurrent set GROUP_M x (32 / GROUP_M) (profiles/r4_pmc_traffic_b8.json:
    // the 256x256 GEGLU tile fetched 335 MB per launch for 79 MB of operands).  Bijective for any tile counts.
    if (p.group_m > 1) {
      const int per = p.group_m * p.ntiles;
      const int gid = lid / per, first = gid * p.group_m;
      const int gsz = p.mtiles - first < p.group_m ? p.mtiles - first : p.group_m;
      const int in = lid - gid * per;
      tm = first + in % gsz;
      tn = in / gsz;
    } else {
      tn = lid % p.ntiles;
      tm = lid / p.ntiles;
    }
  }
  const int M = d.M;
  const u16* __restrict__ Ag = reinterpret_cast<const u16*>(d.A) + (long long)bz * d.strideA_z;
  const u16* __restrict__ Wg = reinterpret_cast<const u16*>(d.W) + (long long)bz * d.strideW_z;

  // ---- staging roles: thread handles LDS chunk position (row = (tid>>3) + RPP*i, cpos = tid&7) ----
  // swizzle key of a tile row: 128-byte rows (BK 64): (row >> 1) & 7 — two rows fill a 256-byte bank line; 64-byte rows
  // (BK 32): (row >> 2) & 3 — four rows per bank line.  Either way the 16 lanes of a ds_read_b128 service group (rows
  // distinct mod 16, same logical chunk) land in 16 distinct 16-byte slots.
  const int srow = (stid < 0 ? 0 : stid) / CPR;
  const int skey = BK == 64 ? ((srow >> 1) & 7) : ((srow >> 2) & 3);
  const int cch = (((stid < 0 ? 0 : stid) % CPR) ^ skey) * 8;  // logical K offset (halfs) of the chunk this thread fetches
  const bool conv = d.mode == DBIR_MODE_CONV3X3;

  // Operands are fetched with `buffer_load_dwordx4 ... lds` through two block-local buffer descriptors (SRD): the
  // per-lane part of an address is a 32-bit byte offset from a block-uniform base (the lowest address the tile can
  // touch), the (tap, channel-tile) part of a K tile goes into the scalar offset operand, and padding / out-of-tile
  // rows use an out-of-range offset for which the hardware returns zeros — no zero page, no 64-bit per-lane pointer
  // math, no select per load (the staging burst was ~1000 of ~2850 cycles per K tile with flat global loads).
  constexpr int OOB = 0x7fffff00;  // >= num_records of every descriptor below
  long long a_ref;                 // element offset (from Ag) of the block's lowest activation address
  {
    const int m0 = tm * BM < M ? tm * BM : M - 1;
    if (conv) {
      const int hw = d.Ho * d.Wo;
      const int b0 = m0 / hw, rem0 = m0 - b0 * hw;
      const int oy0 = rem0 / d.Wo, ox0 = rem0 - oy0 * d.Wo;
      int sy0 = oy0 * d.stride - d.pad, sx0 = ox0 * d.stride - d.pad;
      if (d.upsample) {
        sy0 >>= 1;
        sx0 >>= 1;
      }
      // rows of a tile have non-decreasing (b, oy): reference = start of the first row's source line (sx >= -1);
      // with the nearest-x2 upsample two output lines share a source line, so the column must not enter the bound
      (void)sx0;
      a_ref = ((long long)b0 * d.Hi * d.Wi + (long long)sy0 * d.Wi - 2) * d.Cin;
    } else {
      a_ref = (long long)m0 * d.lda;
    }
  }
  long long a_left = (p.a_elems - a_ref) * 2;  // bytes from the descriptor base to the end of the activation tensor
  if (a_left > 0x7ffffe00LL) a_left = 0x7ffffe00LL;
  if (a_left < 0) a_left = 0;
  const __amdgpu_buffer_rsrc_t a_srd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(Ag) + a_ref, 0, (int)a_left, 0x00020000);
  const long long w_ref = (long long)tn * BN * d.Kpad;
  long long w_left = ((long long)d.Wrows * d.Kpad - w_ref) * 2;
  if (w_left > 0x7ffffe00LL) w_left = 0x7ffffe00LL;
  if (w_left < 0) w_left = 0;
  const __amdgpu_buffer_rsrc_t w_srd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(Wg) + w_ref, 0, (int)w_left, 0x00020000);

  // Activation rows: byte offset (from the descriptor base) of the element of tap (0,0) and a 9-bit tap validity
  // mask; for a nearest-x2 upsampled input bits 16/17 hold the x/y parity that decides whether tap k moves to the
  // next source pixel ((k + parity) >> 1).
  int a_voff[RA];
  unsigned a_mask[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = tm * BM + srow + RPP * i;
    const bool ok = m < M;
    if (conv) {
      const int hw = d.Ho * d.Wo;
      const int mm = ok ? m : 0;
      const int b = mm / hw;
      const int rem = mm - b * hw;
      const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
      const int iy0 = oy * d.stride - d.pad, ix0 = ox * d.stride - d.pad;  // virtual coords of tap (0,0)
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = iy0 + t / 3, ix = ix0 + t % 3;
        if (ok && iy >= 0 && iy < p.Hv && ix >= 0 && ix < p.Wv) mk |= 1u << t;
      }
      int sy = iy0, sx = ix0;
      if (d.upsample) {
        mk |= (unsigned)(ix0 & 1) << 16 | (unsigned)(iy0 & 1) << 17;
        sy >>= 1;  // arithmetic shift: -1 -> -1 (masked)
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
  int w_voff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int nl = srow + RPP * i;
    const bool ok = tn * BN + nl < d.Wrows && nl < BN;
    w_voff[i] = ok ? (int)(((long long)nl * d.Kpad + cch) * 2) : OOB;
  }

  // staging cursor (uniform): K tile, tap, channel tile, ring slot.  K tiles are visited TAP-INNER for convolutions
  // (p.tap_inner): the 9 taps of one 64-channel slice are consecutive, so a block re-reads the same few image rows
  // of that slice from L2 nine times instead of streaming its whole activation panel once per tap (rocprofv3: L2 hit
  // rate 74 % -> see profiles/).  With split-K this block owns K tiles [kt0, kt0 + nk).
  const int kt0 = ksp * p.kt_per;
  int s_kt = kt0, s_slot = 0, s_tap, s_cc;
  if (p.tap_inner) {
    s_cc = kt0 / p.ntaps;
    s_tap = kt0 - s_cc * p.ntaps;
  } else {
    s_tap = kt0 / p.nkc;
    s_cc = kt0 - s_tap * p.nkc;
  }

// issue the direct-to-LDS loads of K tile s_kt into ring slot s_slot, then advance the cursor
#define STAGE()                                                                                     \
  do {                                                                                              \
    char* ab_ = smem + s_slot * BUF_BYTES + swave * 1024;                                           \
    char* bb_ = ab_ + A_BYTES;                                                                      \
    const int ky_ = (s_tap * 11) >> 5, kx_ = s_tap - 3 * ky_;                                       \
    const int koff_ = s_cc * (BK * 2);                          /* bytes, scalar operand */         \
    const unsigned tbit_ = 1u << s_tap;                                                             \
    if (!d.upsample) {                                                                              \
      const int toff_ = (ky_ * d.Wi + kx_) * d.Cin * 2;         /* bytes, uniform */                \
      _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                              \
        const int v_ = (a_mask[i] & tbit_) ? a_voff[i] + toff_ : OOB;                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_srd, (lptr_t)(ab_ + i * PASS_BYTES), 16, v_,     \
                                                 koff_, 0, 0);                                      \
      }                                                                                             \
    } else {                                                                                        \
      _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                              \
        const int dy_ = (ky_ + (int)((a_mask[i] >> 17) & 1)) >> 1;                                  \
        const int dx_ = (kx_ + (int)((a_mask[i] >> 16) & 1)) >> 1;                                  \
        const int v_ = (a_mask[i] & tbit_) ? a_voff[i] + (dy_ * d.Wi + dx_) * d.Cin * 2 : OOB;      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_srd, (lptr_t)(ab_ + i * PASS_BYTES), 16, v_,     \
                                                 koff_, 0, 0);                                      \
      }                                                                                             \
    }                                                                                               \
    const int woff_ = (s_tap * p.nkc + s_cc) * (BK * 2);                                            \
    _Pragma("unroll") for (int i = 0; i < RB; ++i) {                                                \
      if (!EXACT_B || srow + RPP * i < BN) /* lanes past the tile must not write LDS */             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (lptr_t)(bb_ + i * PASS_BYTES), 16,         \
                                                 w_voff[i], woff_, 0, 0);                           \
    }                                                                                               \
    ++s_kt;                                                                                         \
    s_slot = (s_slot + 1 == STAGES) ? 0 : s_slot + 1;                                               \
    if (p.tap_inner) {                                                                              \
      if (++s_tap == p.ntaps) {                                                                     \
        s_tap = 0;                                                                                  \
        ++s_cc;                                                                                     \
      }                                                                                             \
    } else if (++s_cc == p.nkc) {                                                                   \
      s_cc = 0;                                                                                     \
      ++s_tap;                                                                                      \
    }                                                                                               \
  } while (0)

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (p.nkc * p.ntaps - kt0 < p.kt_per) ? p.nkc * p.ntaps - kt0 : p.kt_per;
  if (!LDW || loader) {
#pragma unroll
    for (int s = 0; s < (DEPH ? 1 : STAGES - 1); ++s)
      if (s < nk) STAGE();
  }

  // fragment read offsets (bytes) inside a stage: row * 128 + ((2*ks + hi) ^ key(row)) * 16
  const int a_frag = (wm * 32 * MI + lq) * ROWB;
  const int b_frag = A_BYTES + (wn * 32 * NJ + lq) * ROWB;
  const int sw = BK == 64 ? ((lq >> 1) & 7) : ((lq >> 2) & 3);
  constexpr int FSTR = 32 * ROWB;  // bytes between consecutive 32-row MFMA blocks

  int c_slot = 0;
  if constexpr (LDW > 0) {
    static_assert(!DEPH && !XPF && PIPE == 1 && STAGES >= 3, "producer / consumer tiles: lockstep, pipelined reads, >= 3 slots");
    if (loader) {
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + STAGES - 2 < nk)
          wait_vmcnt<(STAGES - 2) * LOADS>();   // tile kt landed; up to STAGES - 2 later tiles stay in flight
        else
          wait_vmcnt<0>();
        asm volatile("s_barrier" ::: "memory");
        if (kt + STAGES - 1 < nk) STAGE();      // into the slot of tile kt - 1 (its readers passed lgkmcnt(0) + this barrier)
      }
      return;
    }
    typename T::vec8 xf[2][MI], wf[2][NJ];
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const char* base = smem + c_slot * BUF_BYTES;
      c_slot = (c_slot + 1 == STAGES) ? 0 : c_slot + 1;
#define LOAD_FRAGS_P(KS_, SET)                                                                                \
  do {                                                                                                        \
    const int co_ = ((2 * (KS_) + hi) ^ sw) * 16;                                                             \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) wf[SET][j] =                                               \
        *reinterpret_cast<const typename T::vec8*>(base + b_frag + j * FSTR + co_);                           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) xf[SET][i] =                                               \
        *reinterpret_cast<const typename T::vec8*>(base + a_frag + i * FSTR + co_);                           \
  } while (0)
      LOAD_FRAGS_P(0, 0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks < KS - 1) LOAD_FRAGS_P(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma32(wf[ks & 1][j], xf[ks & 1][i], acc[i][j]);  // D[n][m]
      }
#undef LOAD_FRAGS_P
    }
  } else if constexpr (DEPH) {
    static_assert(STAGES == 3 && PIPE == 1 && (WM * WN) % 2 == 0, "DEPH needs a 3-slot ring and two wave groups");
    const int grp = wave >= (WM * WN) / 2 ? 1 : 0;
    wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_barrier" ::: "memory");  // group 1 runs one interval behind group 0
    unsigned long long tacc0 = 0, tacc1 = 0, tacc2 = 0, tacc3 = 0, tprev = 0;
    const bool instr = kDiag && p.debug == 5;
#define TSD(ACC)                                                   \
  do {                                                             \
    if (instr) {                                                   \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      ACC += now_ - tprev;                                         \
      tprev = now_;                                                \
    }                                                              \
  } while (0)
    if (instr) tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) STAGE();
      if (grp == 0) {
        if (more)
          wait_vmcnt<LOADS>();
        else
          wait_vmcnt<0>();
      }
      TSD(tacc0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      TSD(tacc1);
      const char* base = smem + c_slot * BUF_BYTES;
      c_slot = (c_slot + 1 == STAGES) ? 0 : c_slot + 1;
      {
        typename T::vec8 xf[2][MI], wf[2][NJ];
#define LOAD_FRAGS_D(KS, SET)                                                                                 \
  do {                                                                                                        \
    const int co_ = ((2 * (KS) + hi) ^ sw) * 16;                                                              \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) wf[SET][j] =                                               \
        *reinterpret_cast<const typename T::vec8*>(base + b_frag + j * FSTR + co_);                           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) xf[SET][i] =                                               \
        *reinterpret_cast<const typename T::vec8*>(base + a_frag + i * FSTR + co_);                           \
  } while (0)
        LOAD_FRAGS_D(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (ks < KS - 1) LOAD_FRAGS_D(ks + 1, (ks + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma32(wf[ks & 1][j], xf[ks & 1][i], acc[i][j]);
        }
#undef LOAD_FRAGS_D
      }
      if (grp == 1) wait_vmcnt<0>();
      TSD(tacc2);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      TSD(tacc3);
    }
    if (instr && p.ws && lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws) + ((long long)blockIdx.x * (WM * WN) + wave) * 8;
      o[0] = tacc0; o[1] = tacc1; o[2] = tacc2; o[3] = tacc3;
      o[4] = __builtin_amdgcn_s_memtime() - tstart; o[5] = nk; o[6] = grp;
    }
#undef TSD
    if (grp == 0) asm volatile("s_barrier" ::: "memory");  // pairs group 1's extra barrier
  } else if constexpr (XPF) {
    static_assert(PIPE == 1 && (KS % 2) == 0, "cross-tile prefetch needs pipelined fragment reads and an even k-step count");
    typename T::vec8 xf[2][MI], wf[2][NJ];
    const char* base;
#define LOAD_FRAGS_X(KS_, SET)                                                                                \
  do {                                                                                                        \
    const int co_ = ((2 * (KS_) + hi) ^ sw) * 16;                                                             \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) wf[SET][j] =                                               \
        *reinterpret_cast<const typename T::vec8*>(base + b_frag + j * FSTR + co_);                           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) xf[SET][i] =                                               \
        *reinterpret_cast<const typename T::vec8*>(base + a_frag + i * FSTR + co_);                           \
  } while (0)
// tile KT_ landed (this wave's share; up to STAGES - 2 later tiles stay in flight), every wave's share landed and every
// wave has COMPLETED its reads of the previous tile (lgkmcnt(0)): that slot is refilled right behind the barrier
#define ACQUIRE_X(KT_)                                                                                        \
  do {                                                                                                        \
    if ((KT_) + STAGES - 2 < nk) wait_vmcnt<(STAGES - 2) * LOADS>(); else wait_vmcnt<0>();                    \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                          \
    base = smem + c_slot * BUF_BYTES;                                                                         \
    c_slot = (c_slot + 1 == STAGES) ? 0 : c_slot + 1;                                                         \
  } while (0)
    ACQUIRE_X(0);
    LOAD_FRAGS_X(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (STAGES - 1 < nk) STAGE();
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks < KS - 1) {
          LOAD_FRAGS_X(ks + 1, (ks + 1) & 1);
        } else if (kt + 1 < nk) {
          ACQUIRE_X(kt + 1);
          LOAD_FRAGS_X(0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (kt + STAGES < nk) STAGE();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma32(wf[ks & 1][j], xf[ks & 1][i], acc[i][j]);  // D[n][m]
      }
    }
#undef LOAD_FRAGS_X
#undef ACQUIRE_X
  } else {
  // DIAGNOSTIC (kDiag build, debug == 5): per-wave s_memtime accumulators {vmcnt wait, barrier wait, first reads +
  // stage issue, remaining reads + MFMA issue} -> workspace, read by tools/probes/bench_one.py
  unsigned long long tacc0 = 0, tacc1 = 0, tacc2 = 0, tacc3 = 0, tprev = 0;
  const bool instr = kDiag && p.debug == 5;
#define TSL(ACC)                                                   \
  do {                                                             \
    if (instr) {                                                   \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      ACC += now_ - tprev;                                         \
      tprev = now_;                                                \
    }                                                              \
  } while (0)
  if (instr) tprev = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = tprev;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt landed (this wave's share); later tiles may stay in flight
    if (kt + STAGES - 2 < nk)
      wait_vmcnt<(STAGES - 2) * LOADS>();
    else
      wait_vmcnt<0>();
    TSL(tacc0);
    // every wave's share landed, and everyone is done reading the slot that is refilled next
    asm volatile("s_barrier" ::: "memory");
    TSL(tacc1);
    const char* base = smem + c_slot * BUF_BYTES;
    c_slot = (c_slot + 1 == STAGES) ? 0 : c_slot + 1;
    if constexpr (!PIPE) {
      if (kt + STAGES - 1 < nk && (!kDiag || p.debug != 1)) STAGE();
      TSL(tacc2);
    }
    if constexpr (PIPE) {
      typename T::vec8 xf[2][MI], wf[2][NJ];
#define LOAD_FRAGS(KS, SET)                                                                                   \
  do {                                                                                                        \
    const int co_ = ((2 * (KS) + hi) ^ sw) * 16;                                                              \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) wf[SET][j] =                                               \
        *reinterpret_cast<const typename T::vec8*>(base + b_frag + j * FSTR + co_);                           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) xf[SET][i] =                                               \
        *reinterpret_cast<const typename T::vec8*>(base + a_frag + i * FSTR + co_);                           \
  } while (0)
      LOAD_FRAGS(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the next tile's direct-to-LDS loads are issued while the first fragment reads are in flight (the slot they
      // refill was last read before the barrier above)
      if (kt + STAGES - 1 < nk && (!kDiag || p.debug != 1)) STAGE();
      TSL(tacc2);
      if (kDiag && p.debug == 2) continue;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks < KS - 1) LOAD_FRAGS(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);  // keep the next k-step's reads ahead of this k-step's MFMAs
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma32(wf[ks & 1][j], xf[ks & 1][i], acc[i][j]);  // D[n][m]
      }
#undef LOAD_FRAGS
    } else {
      if (kDiag && p.debug == 2) continue;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int co = ((2 * ks + hi) ^ sw) * 16;
        typename T::vec8 xf[MI], wf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          wf[j] = *reinterpret_cast<const typename T::vec8*>(base + b_frag + j * FSTR + co);
#pragma unroll
        for (int i = 0; i < MI; ++i)
          xf[i] = *reinterpret_cast<const typename T::vec8*>(base + a_frag + i * FSTR + co);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma32(wf[j], xf[i], acc[i][j]);  // D[n][m]
      }
    }
    TSL(tacc3);
  }
  if (instr && p.ws && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws) + ((long long)blockIdx.x * (WM * WN) + wave) * 8;
    o[0] = tacc0; o[1] = tacc1; o[2] = tacc2; o[3] = tacc3;
    o[4] = __builtin_amdgcn_s_memtime() - tstart; o[5] = nk; o[6] = 0;
  }
#undef TSL
  }  // lockstep path
#undef STAGE

  {
    EpiParams ep;
    ep.vec_bias = p.vec_bias;
    ep.vec_rv = p.vec_rv;
    ep.splitk = p.splitk;
    ep.ws = p.ws;
    gemm_epilogue<T, WM, WN, MI, NJ>(d, ep, acc, smem, tm, tn, ksp, bz, wm, wn, tid);
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// Split-K second pass: sum the K slices in a fixed order (deterministic), then the same epilogue as the fused path
// (bias, row vector, activation, scale, residual) and the 16-bit store.  One thread = 8 consecutive columns.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const G2Params p) {
  const dbir_gemm_desc& d = p.d;
  const int M = d.M, N = d.N, n8 = N >> 3;
  const long long total = (long long)M * n8;
  const int bz = blockIdx.y;
  const float* __restrict__ wsp = p.ws + (long long)bz * p.splitk * (long long)M * N;
  const u16* __restrict__ RV = reinterpret_cast<const u16*>(d.rowvec);
  const u16* __restrict__ Rg = d.R ? reinterpret_cast<const u16*>(d.R) + (long long)bz * d.strideR_z : nullptr;
  u16* __restrict__ Cg = reinterpret_cast<u16*>(d.C) + (long long)bz * d.strideC_z;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int m = (int)(q / n8), n0 = (int)(q - (long long)m * n8) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int sidx = 0; sidx < p.splitk; ++sidx) {
      const float4* src = reinterpret_cast<const float4*>(wsp + ((long long)sidx * M + m) * N + n0);
      const float4 a = src[0], b = src[1];
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    if (d.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += d.bias[n0 + e];
    }
    if (RV) {
      const u16* rvp = RV + (long long)(m / d.rows_per_batch) * d.rowvec_ld + n0;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += T::to_f32(rvp[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e];
      if (d.act == DBIR_ACT_SILU) x = silu_f(x);
      else if (d.act == DBIR_ACT_GELU) x = gelu_fast(x);
      else if (d.act == DBIR_ACT_LRELU) x = x > 0.f ? x : x * d.act_param;
      // same rounding points as the fused epilogue: round to 16 bit, then add the residual in f32, round again
      v[e] = T::to_f32(T::from_f32(x * d.out_scale));
    }
    if (Rg) {
      const uint4 rr = *reinterpret_cast<const uint4*>(Rg + (long long)m * d.ldr + n0);
      float b8[8];
      unpack8<T>(rr, b8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += b8[e];
    }
    *reinterpret_cast<uint4*>(Cg + (long long)m * d.ldc + n0) = pack8<T>(v);
  }
}

// The same second pass when the caller asked for GroupNorm statistics of the output (dbir_gemm_desc.stats; round 4: split-K
// launches used to fall back to a separate statistics kernel over the stored tensor): a block owns a 64-row stripe x 64
// columns (grid M / 64 x N / 64: the problems that need split-K are the small ones, they need the blocks), thread (row lane
// rl = tid / 8, chunk cl = tid % 8) handles rows rl and rl + 32 of its 8-column chunk, so the column statistics of the
// STORED values (shifted sums, merged pairwise in a fixed order: gemm_epilogue.h ColStat) come out per 64-row tile:
// stats[m / 64][2][N].  Needs M % 64 == 0 and one z slice.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const G2Params p) {
  __shared__ __attribute__((aligned(16))) char sm[32 * 8 * 68];
  const dbir_gemm_desc& d = p.d;
  const int M = d.M, N = d.N;
  const int tid = threadIdx.x, rl = tid >> 3, cl = tid & 7;
  const int n0 = (blockIdx.y * 8 + cl) * 8;
  const float* __restrict__ wsp = p.ws;
  const u16* __restrict__ RV = reinterpret_cast<const u16*>(d.rowvec);
  const u16* __restrict__ Rg = d.R ? reinterpret_cast<const u16*>(d.R) : nullptr;
  u16* __restrict__ Cg = reinterpret_cast<u16*>(d.C);
  ColStat cs;
  colstat_init(cs);
  const bool active = n0 < N;
  if (active) {
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = d.bias ? d.bias[n0 + e] : 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int m = blockIdx.x * 64 + rl + 32 * k;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      for (int sidx = 0; sidx < p.splitk; ++sidx) {
        const float4* src = reinterpret_cast<const float4*>(wsp + ((long long)sidx * M + m) * N + n0);
        const float4 a = src[0], b = src[1];
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias8[e];
      if (RV) {
        const u16* rvp = RV + (long long)(m / d.rows_per_batch) * d.rowvec_ld + n0;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += T::to_f32(rvp[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[e];
        if (d.act == DBIR_ACT_SILU) x = silu_f(x);
        else if (d.act == DBIR_ACT_GELU) x = gelu_fast(x);
        else if (d.act == DBIR_ACT_LRELU) x = x > 0.f ? x : x * d.act_param;
        v[e] = T::to_f32(T::from_f32(x * d.out_scale));   // same rounding points as the fused epilogue
      }
      if (Rg) {
        const uint4 rr = *reinterpret_cast<const uint4*>(Rg + (long long)m * d.ldr + n0);
        float b8[8];
        unpack8<T>(rr, b8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b8[e];
      }
      const uint4 pk = pack8<T>(v);
      *reinterpret_cast<uint4*>(Cg + (long long)m * d.ldc + n0) = pk;
      unpack8<T>(pk, v);   // statistics of what was stored
      colstat_add(cs, v);
    }
  }
  colstat_finish<256, 64, 8, 32>(cs, active, sm, tid, cl, rl, d.stats + (long long)blockIdx.x * 2 * N, blockIdx.y, N);
}

// launch the split-K second pass (with or without the statistics stage) for the f32 slab layout [z][slice][M][N]
template <typename T>
int splitk_second_pass(const G2Params& p, hipStream_t s) {
  const unsigned nz = p.d.batch > 0 ? p.d.batch : 1;
  if (p.d.stats) {   // (callers only leave stats set when M % 64 == 0 and nz == 1)
    hipLaunchKernelGGL((splitk_reduce_stats_kernel<T>), dim3((unsigned)(p.d.M / 64), (unsigned)cdiv(p.d.N >> 3, 8)), dim3(256),
                       0, s, p);
  } else {
    const long long work = (long long)p.d.M * (p.d.N >> 3);
    const unsigned rb = (unsigned)(work / 256 + 1 < 4096 ? work / 256 + 1 : 4096);
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(rb, nz), dim3(256), 0, s, p);
  }
  DBIR_CHECK_LAUNCH("dbir_gemm(split-K reduce)");
  return DBIR_OK;
}

template <typename T, int WM, int WN, int MI, int NJ, int STAGES, int PIPE = 0, int DEPH = 0, int BKT = 64, int WPS = 0,
          int XPF = 0, int LDW = 0>
int launch2(G2Params& p, hipStream_t s) {
  constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
  constexpr int RPP_ = 64 * (LDW ? LDW : WM * WN) / (BKT / 8), BNR = DEPH ? BN : (BN + RPP_ - 1) / RPP_ * RPP_;
  constexpr int ring = STAGES * (BM + BNR) * BKT * 2, epi = BM * (BN + 8) * 2;  // operand ring / transposed C tile
  constexpr int lds = ring > epi ? ring : epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  constexpr int blocks_per_cu = (160 * 1024) / lds;
  constexpr int waves = (WM * WN + LDW) * (blocks_per_cu > 2 ? 2 : blocks_per_cu);
  // WPS > 0: explicit waves per SIMD (register budget 512 / WPS) for variants meant to run several blocks per CU
  constexpr int MINW = WPS > 0 ? WPS : (waves > 8 ? 3 : (waves >= 8 ? 2 : 1));
  auto kern = &gemm_glds_kernel<T, WM, WN, MI, NJ, STAGES, MINW, PIPE, DEPH, BKT, XPF, LDW>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const dbir_gemm_desc& dd = p.d;
  // K tiling of this variant (K / Cin are multiples of 64, checked by dbir_gemm_glds_eligible) and the split-K slices:
  // whole K tiles, every slice non-empty
  p.nkc = (dd.mode == DBIR_MODE_LINEAR ? dd.K : dd.Cin) / BKT;
  {
    const int nk_total = p.nkc * p.ntaps;
    int sk = dd.splitk > 1 ? dd.splitk : 1;
    if (sk > nk_total) sk = nk_total;
    p.kt_per = cdiv(nk_total, sk);
    p.splitk = cdiv(nk_total, p.kt_per);
    if (p.debug == 5 && p.splitk > 1) p.debug = 0;  // the diagnostic timestamps share the workspace pointer
    if (p.splitk > 1) {
      const long long need = (long long)p.splitk * (dd.batch > 0 ? dd.batch : 1) * dd.M * dd.N * 4;
      if (!dd.ws || dd.ws_bytes < need || (reinterpret_cast<uintptr_t>(dd.ws) & 15)) {
        dbir_set_error("dbir_gemm: split-K %d needs a 16-byte aligned workspace of %lld bytes (got %lld)", p.splitk,
                       need, dd.ws_bytes);
        return DBIR_ERR_ARG;
      }
      if (dd.act == DBIR_ACT_GEGLU || dd.N % 8 != 0 || dd.ldc % 8 != 0) {
        dbir_set_error("dbir_gemm: split-K needs N %% 8 == 0 and no GEGLU");
        return DBIR_ERR_ARG;
      }
    }
  }
  p.mtiles = cdiv(p.d.M, BM);
  p.ntiles = cdiv(p.d.N, BN);
  {
    // OFF by default: measured per shape in the two-stream evaluation (profiles/r4_group_m_ab.txt) GROUP_M = 4 is neutral
    // on the wide GEGLU tiles it was meant for and costs the 128x128 tile 2.4 us (6 %) on 4096 x 1280 x 1280; the L2-miss
    // traffic it removes is not what those launches wait for.  DBIR_GROUP_M=4 switches it on for experiments.
    static const int gm_env = getenv("DBIR_GROUP_M") ? atoi(getenv("DBIR_GROUP_M")) : 1;
    p.group_m = (gm_env > 1 && p.ntiles >= 8 && p.mtiles >= gm_env) ? gm_env : 1;
  }
  if (p.d.stats) {  // GroupNorm column statistics of the output: from the epilogue (whole tiles only), or — under split-K —
                    // from the reduce pass in 64-row tiles
    if (p.splitk <= 1 && p.d.M % BM == 0) g_dbir_stats_rows = BM;
    else if (p.splitk > 1 && p.d.M % 64 == 0 && p.d.batch <= 1) g_dbir_stats_rows = 64;
    else p.d.stats = nullptr;
  }
  const unsigned nz = p.d.batch > 0 ? p.d.batch : 1;
  dim3 grid((unsigned)(p.mtiles * p.ntiles * p.splitk), nz);
  if (LDW && p.splitk > 1) {
    dbir_set_error("dbir_gemm: the producer / consumer tiles (90 - 92) do not do split-K");
    return DBIR_ERR_ARG;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * (WM * WN + LDW)), lds, s, p);
  DBIR_CHECK_LAUNCH("dbir_gemm(glds)");
  if (p.splitk > 1) return splitk_second_pass<T>(p, s);
  return DBIR_OK;
}

template <typename T>
int dispatch2(G2Params& p, int tile, hipStream_t s) {
  switch (tile) {
    case 5: return launch2<T, 2, 2, 2, 2, 2>(p, s);   // 128x128, 4 waves, 2 stages, 2 blocks / CU
    case 6: return launch2<T, 4, 1, 2, 2, 2>(p, s);   // 256x64
    case 7: return launch2<T, 1, 4, 2, 2, 2>(p, s);   // 64x256
    case 8: return launch2<T, 2, 2, 4, 2, 3>(p, s);   // 256x128, 4 waves (128x64 each), 3 stages
    case 9: return launch2<T, 4, 2, 2, 2, 3>(p, s);   // 256x128, 8 waves (64x64 each), 3 stages
    case 10: return launch2<T, 2, 4, 4, 2, 2>(p, s);  // 256x256, 8 waves (128x64 each), 2 stages
    case 11: return launch2<T, 2, 2, 2, 2, 4>(p, s);  // 128x128, 4 waves, 4 stages, 1 block / CU
    case 12: return launch2<T, 4, 2, 2, 2, 2>(p, s);  // 256x128, 8 waves, 2 stages
    case 14: return launch2<T, 8, 1, 1, 5, 2>(p, s);  // 256x160, 8 waves (32x160 each): N = 320 k without padding
    case 15: return launch2<T, 4, 1, 1, 5, 2>(p, s);  // 128x160, 4 waves, 2 blocks / CU
    case 16: return launch2<T, 4, 1, 2, 5, 2>(p, s);  // 256x160, 4 waves (64x160 each)
    // 20 + t: tile t with software-pipelined fragment reads (PIPE = 1)
    case 25: return launch2<T, 2, 2, 2, 2, 2, 1>(p, s);
    case 26: return launch2<T, 4, 1, 2, 2, 2, 1>(p, s);
    case 30: return launch2<T, 2, 4, 4, 2, 2, 1>(p, s);
    case 32: return launch2<T, 4, 2, 2, 2, 2, 1>(p, s);
    case 34: return launch2<T, 8, 1, 1, 5, 2, 1>(p, s);
    case 35: return launch2<T, 4, 1, 1, 5, 2, 1>(p, s);
    // de-phased two-group variants (3-slot ring): staging bursts of one group overlap the MFMAs of the other
    case 36: return launch2<T, 4, 2, 2, 2, 3, 1, 1>(p, s);  // 256x128
    case 37: return launch2<T, 8, 1, 1, 5, 3, 1, 1>(p, s);  // 256x160
    case 38: return launch2<T, 4, 2, 2, 1, 3, 1, 1>(p, s);  // 256x64
    // K depth 32: the 256x256 tile fits a 3-slot ring (96 KB) -> de-phased / 4-slot lockstep variants
    case 40: return launch2<T, 2, 4, 4, 2, 3, 1, 1, 32>(p, s);  // 256x256 de-phased
    case 41: return launch2<T, 2, 4, 4, 2, 4, 1, 0, 32>(p, s);  // 256x256 lockstep, 4-slot ring
    // K depth 32, <= 80 KB of LDS and <= 128 VGPRs: TWO (three) independent blocks per CU, so one block's prologue /
    // epilogue (HBM-bound, GELU-heavy for GEGLU) overlaps another block's K loop without any in-kernel scheduling
    case 44: return launch2<T, 4, 2, 2, 2, 3, 1, 0, 32, 4>(p, s);  // 256x128, 8 waves (64x64 each), 3-slot ring, 72 KB
    case 45: return launch2<T, 2, 2, 2, 2, 3, 1, 0, 32, 3>(p, s);  // 128x128, 4 waves, 3-slot ring, 48 KB: 3 blocks / CU
    // (80 - 88, round 3: the lockstep tiles with cross-tile fragment prefetch, XPF = 1 — 0.98 - 1.02x the incumbents on every
    // shape, profiles/r3_autotune_xprefetch_tiles_b8.log — are no longer instantiated; the schedule stays in the kernel
    // template as the reference for the fused transformer kernels' tile loop, which uses it)
    // 90 - 92: producer / consumer split (LDW = 4 loader waves + the matrix waves), 3-slot ring (a 256x160 variant with 8
    // matrix waves needs > 168 VGPRs at three waves per SIMD and spilled: measured slower everywhere, not kept)
    case 90: return launch2<T, 4, 1, 1, 5, 3, 1, 0, 64, 0, 0, 4>(p, s);   // 128x160, 4 matrix waves (32x160 each)
    case 91: return launch2<T, 2, 2, 2, 2, 3, 1, 0, 64, 0, 0, 4>(p, s);   // 128x128, 4 matrix waves (64x64 each)
    case 92: return launch2<T, 4, 2, 2, 2, 3, 1, 0, 64, 0, 0, 4>(p, s);   // 256x128, 8 matrix waves (64x64 each)
  }
  dbir_set_error("dbir_gemm: bad glds tile %d", tile);
  return DBIR_ERR_ARG;
}

}  // namespace

// split-K second pass for other kernels that write the same f32 slab layout (gemm_halo.hip)
int dbir_splitk_reduce_launch(const dbir_gemm_desc& d, int splitk, float* ws, hipStream_t s) {
  G2Params p;
  memset(&p, 0, sizeof(p));
  p.d = d;
  p.splitk = splitk;
  p.ws = ws;
  return d.dtype == DBIR_F16 ? splitk_second_pass<F16>(p, s) : splitk_second_pass<BF16>(p, s);
}

// Is the descriptor (already validated by dbir_gemm) runnable on the direct-to-LDS kernel?
bool dbir_gemm_glds_eligible(const dbir_gemm_desc& d) {
  if (d.out_f32) {  // direct float4 stores from the accumulators: plain epilogues only
    if (d.store_mode != 0 || d.R || d.act == DBIR_ACT_GEGLU || d.splitk > 1 || d.ldc % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(d.C) & 15) || d.strideC_z % 4 != 0)
      return false;
    if ((reinterpret_cast<uintptr_t>(d.A) & 15) || (reinterpret_cast<uintptr_t>(d.W) & 15)) return false;
    if (d.strideA_z % 8 || d.strideW_z % 8) return false;
    if (d.mode == DBIR_MODE_LINEAR) return d.K % BK == 0 && d.lda % 8 == 0;
    return d.Cin % BK == 0 && (long long)d.B * d.Hi * d.Wi < 2147483647LL;
  }
  if (d.store_mode == 1) {  // transposed store: whole 8-row chunks inside one batch, 16-byte aligned destinations
    if (d.trans_L % 8 != 0 || d.trans_ld % 8 != 0 || d.trans_bstride % 8 != 0 || d.R || d.act == DBIR_ACT_GEGLU ||
        d.splitk > 1 || d.batch > 1)
      return false;
  } else if (d.store_mode != 0) {
    return false;
  }
  if (d.ldc % 8 != 0 || (reinterpret_cast<uintptr_t>(d.C) & 15)) return false;
  if (d.R && (d.ldr % 8 != 0 || (reinterpret_cast<uintptr_t>(d.R) & 15))) return false;
  if ((reinterpret_cast<uintptr_t>(d.A) & 15) || (reinterpret_cast<uintptr_t>(d.W) & 15)) return false;
  if (d.strideA_z % 8 || d.strideW_z % 8 || d.strideC_z % 8 || d.strideR_z % 8) return false;
  if (d.mode == DBIR_MODE_LINEAR) {
    if (d.K % BK != 0 || d.lda % 8 != 0) return false;
  } else {
    if (d.Cin % BK != 0) return false;
    if ((long long)d.B * d.Hi * d.Wi >= 2147483647LL) return false;
  }
  if (d.act == DBIR_ACT_GEGLU && d.N % 64 != 0) return false;
  return true;
}

int dbir_gemm_glds(const dbir_gemm_desc& dd, int Hv, int Wv, int tile, hipStream_t s) {
  G2Params p;
  p.d = dd;
  p.Hv = Hv;
  p.Wv = Wv;
  p.ntaps = dd.mode == DBIR_MODE_LINEAR ? 1 : 9;
  p.nkc = 0;  // set per variant in launch2 (depends on the variant's K depth)
  p.vec_bias = dd.bias && (reinterpret_cast<uintptr_t>(dd.bias) & 15) == 0;
  p.vec_rv = dd.rowvec && (reinterpret_cast<uintptr_t>(dd.rowvec) & 7) == 0 && dd.rowvec_ld % 4 == 0;
  {
    static const int dbg = getenv("DBIR_GEMM_DEBUG") ? atoi(getenv("DBIR_GEMM_DEBUG")) : 0;
    static const int tap_inner = getenv("DBIR_TAP_INNER") ? atoi(getenv("DBIR_TAP_INNER")) : 1;  // A/B switch
    p.debug = dbg;
    p.tap_inner = tap_inner;
    p.a_elems = dd.mode == DBIR_MODE_CONV3X3 ? (long long)dd.B * dd.Hi * dd.Wi * dd.Cin
                                             : (long long)(dd.M - 1) * dd.lda + dd.K;
  }
  p.ws = reinterpret_cast<float*>(dd.ws);
  p.splitk = 1;
  p.kt_per = 0;
  if (tile == 0) {
    // Tile choice from the MI355X microbenchmarks (tools/bench_kernels.py, profiles/kbench_r1.json):
    //   256x256 (tile 10, 128x64 per wave: fewest LDS reads per MFMA) whenever N wastes <= 20 % of 256-wide column
    //   tiles and there are enough tiles to occupy the 256 CUs; 256x128 (tile 12) for large M otherwise;
    //   128x128 at 2 blocks / CU (tile 5) for the small-M (8x8 latent level, text context) problems.
    const int nt256 = cdiv(dd.N, 256);
    const bool fits256 = (nt256 * 256 - dd.N) * 5 <= nt256 * 256;
    const long long blocks256 = (long long)cdiv(dd.M, 256) * nt256 * (dd.batch > 0 ? dd.batch : 1);
    if (dd.act == DBIR_ACT_GEGLU || (fits256 && blocks256 >= 150))
      tile = 10;
    else if (dd.M >= 4096)
      tile = 12;
    else
      tile = 5;
  }
  if (dd.act == DBIR_ACT_GEGLU && (tile == 14 || tile == 15 || tile == 16 || tile == 34 || tile == 35 || tile == 37 || tile == 38 ||
                                   tile == 90)) {
    dbir_set_error("dbir_gemm: GEGLU needs a tile whose waves hold value/gate column pairs (tiles 5-13)");
    return DBIR_ERR_ARG;
  }
  return dd.dtype == DBIR_F16 ? dispatch2<F16>(p, tile, s) : dispatch2<BF16>(p, tile, s);
}
