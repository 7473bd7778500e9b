// 3x3 convolution (stride 1, pad 1) as implicit GEMM on gfx950 MFMA with an LDS-RESIDENT HALO PATCH.
//
// Why (profiles/r1_pmc_gemm_probe.txt): in gemm_glds.hip every K tile (tap, 64-channel slice) re-stages a
// BM x 64 activation tile although the nine shifted windows of one channel slice overlap almost completely; the K loop
// is paced by operand delivery (the vector memory pipe moves ~57 B/clk/CU: 53 KB per K tile of the 256x160 tile =
// ~930 of the ~1280 matrix-pipe cycles), not by the matrix pipe.  Here the activations of one channel slice are staged
// ONCE as a patch [rows + 2 halo rows][Wo][64 ch] and the MFMA A-fragments of all 9 taps are read from it at a per-tap
// LDS offset: activation traffic through the vector memory pipe drops from 9 x 32 KB to <= 48 KB per slice (total
// A + W per K tile 53 KB -> ~26 KB for the 256x160 tile), as do the LDS write traffic and the staging instructions
// (<= 4 direct-to-LDS loads per wave and K tile instead of 7).
//
// Geometry.  A block owns BM consecutive output pixels m = (b*Ho + oy)*Wo + ox with BM % Wo == 0, i.e. R = BM / Wo whole
// image rows: either inside one image (Ho % R == 0: one segment of R rows) or R / Ho whole images (R % Ho == 0: one
// segment of Ho rows per image).  Patch = per segment (rows + 2) x Wo pixels (the rows above / below are real
// neighbours or, at an image border / past the batch, zeros delivered by out-of-range buffer loads); the x halo is not
// stored: every tile spans full rows, so x = -1 / x = Wo are always padding and those lanes read a shared zero pixel.
// Pixel p of the patch lives at byte p*128 of the patch buffer, its 8 16-byte chunks XOR-swizzled with key (p>>1)&7 on
// the SOURCE side of the direct-to-LDS load (lane-linear LDS image) and on the ds_read_b128 — 16 consecutive pixels hit
// 16 distinct 16-byte slots whatever the tap shift, so the fragment reads stay conflict-free.
//
// Schedule (same two-group de-phased structure as gemm_glds.hip DEPH, re-derived for the patch): K tiles are visited
// slice-outer / tap-inner; weights go through a 3-slot ring, the patch is double-buffered.  In K tile (c, t) a wave
// issues the weight tile of the NEXT K tile and, for 1 <= t <= NPASS, piece t-1 of patch(c+1); group 0 then waits with
// a counted vmcnt (only the loads of that last stage may stay in flight), group 1 (one interval behind) drains after
// its multiply; two workgroup barriers per K tile.  Hazards:
//   RAW weights: as gemm_glds.hip DEPH.  RAW patch: its pieces are older than the weight tile of (c+1, 0) in each
//       wave's in-order vmcnt queue, so the wait that retires that weight tile retires them.
//   WAR patch: buffer (c+1)&1 was last read by group 1 in K tile (c-1, 8), which ends before the barrier that opens
//       group 0's K tile (c, 1) — the first one that refills it (hence pieces start at t = 1, not t = 0).
// Split-K slices the CHANNEL SLICES (each with all 9 taps); partial sums / reduce kernel shared with gemm_glds.hip.
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"

extern thread_local int g_dbir_stats_rows;  // gemm.hip

namespace {

struct HParams {
  dbir_gemm_desc d;
  int mtiles, ntiles;
  int nsl;              // channel slices (Cin / 64)
  int splitk, sl_per;   // split-K: slices (1 = off), channel slices per K slice
  int vec_bias, vec_rv;
  float* ws;
  int log2wo, log2rps;  // Wo and rows-per-segment are powers of two
  int group_m;           // tile order: GROUP_M row tiles x all column tiles per group (1 = column tiles fastest)
  int nseg, P;          // segments per tile, patch pixels actually used (<= PMAX)
  long long a_elems;
};

typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ABL (diagnostic ablations, tiles 60-63, timing only — results are wrong by construction): 1 = no epilogue, 2 = no
// MFMAs, 3 = no steady-state staging, 4 = no fragment reads.
// LS = 1: lockstep schedule with cross-tile fragment prefetch (see the LS block in the body) instead of the two
// de-phased wave groups.
template <typename T, int WM, int WN, int MI, int NJ, int PMAX, int ABL = 0, int LS = 0>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_halo_kernel(const HParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
  constexpr int RPP = NT / 8;                       // rows / pixels covered by one pass of the whole block
  constexpr int PASS_BYTES = NT * 16;
  constexpr int NPASS = (PMAX + RPP - 1) / RPP;     // patch pieces per channel slice
  constexpr int RB = (BN + RPP - 1) / RPP;          // weight passes per K tile
  constexpr bool EXACT_B = (BN % RPP) != 0;         // last weight pass only partly inside the tile
  constexpr int RB0 = RB;                           // loads per weight stage of a group-0 wave (holds the low rows)
  constexpr int PBUF_BYTES = PMAX * 128 + 256;      // patch + one zero pixel (kept 256-byte aligned)
  constexpr int B_BYTES = BN * 128;
  constexpr int B_BASE = 2 * PBUF_BYTES;
  constexpr int STAGES = 3;
  static_assert(PMAX % RPP == 0, "patch capacity must be whole passes");
  static_assert(NPASS <= 8, "patch pieces are issued in taps 1..8");
  static_assert((WM * WN) % 2 == 0, "two wave groups");
  static_assert(!EXACT_B || (BN % RPP) * 2 == RPP, "partial weight pass must be exactly the group-0 half");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const dbir_gemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lq = lane & 31, hi = lane >> 5;
  const int bz = blockIdx.y;
  const int grp = wave >= (WM * WN) / 2 ? 1 : 0;

  // ---- XCD-aware tile mapping (bijective), as gemm_glds.hip ----
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
  const int Wo = d.Wo, Ho = d.Ho;
  const int rps = 1 << p.log2rps;
  const u16* __restrict__ Ag = reinterpret_cast<const u16*>(d.A) + (long long)bz * d.strideA_z;
  const u16* __restrict__ Wg = reinterpret_cast<const u16*>(d.W) + (long long)bz * d.strideW_z;

  // ---- tile origin: first global image row g0 = b0*Ho + y0 ----
  const int g0 = (tm * BM) >> p.log2wo;
  const int b0 = g0 / Ho, y0 = g0 - b0 * Ho;

  // ---- buffer descriptors (block-uniform) ----
  constexpr int OOB = 0x7fffff00;
  const long long a_ref = ((long long)g0 - 1) * Wo * d.Cin;  // first halo row (may lie before the tensor: never read)
  long long a_left = (p.a_elems - a_ref) * 2;
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

  // ---- staging roles: thread -> (pixel / weight row srow + RPP*i, LDS chunk position tid & 7) ----
  const int srow = tid >> 3;
  const int skey = (srow >> 1) & 7;              // RPP is a multiple of 16: the key does not depend on the pass
  const int cch = ((tid & 7) ^ skey) * 8;        // logical channel offset (halfs) of the chunk this thread fetches
  int a_voff[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int pp = srow + RPP * i;
    const int prow = pp >> p.log2wo, x = pp & (Wo - 1);
    const int s = prow / (rps + 2), r = prow - s * (rps + 2) - 1;
    const int b = b0 + s, y = y0 + r;
    const bool ok = pp < p.P && b < d.B && y >= 0 && y < Ho;
    // element offset from a_ref: halo-row-relative row index (b0*Ho + y0 - 1 is row 0 of the descriptor)
    const long long e = (((long long)(b - b0) * Ho + (y - y0 + 1)) * Wo + x) * d.Cin + cch;
    a_voff[i] = ok ? (int)(e * 2) : OOB;
  }
  int w_voff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int nl = srow + RPP * i;
    const bool ok = tn * BN + nl < d.Wrows && nl < BN;
    w_voff[i] = ok ? (int)(((long long)nl * d.Kpad + cch) * 2) : OOB;
  }

  // ---- fragment roles ----
  // activation block i of this wave: output pixel -> centre pixel of the patch, + left / right border flags
  int ppc[MI];
  bool xl[MI], xr[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int mrow = wm * 32 * MI + i * 32 + lq;
    const int lr = mrow >> p.log2wo, ox = mrow & (Wo - 1);
    const int s = lr >> p.log2rps, r = lr & (rps - 1);
    ppc[i] = ((s * (rps + 2) + r + 1) << p.log2wo) + ox;
    xl[i] = ox == 0;
    xr[i] = ox == Wo - 1;
  }
  const int b_frag = B_BASE + (wn * 32 * NJ + lq) * 128;
  const int sw = (lq >> 1) & 7;
  constexpr int FSTR = 32 * 128;
  const int hik = hi * 16;

  // zero pixels (one per patch buffer)
  if (tid < 16) *reinterpret_cast<uint4*>(smem + PMAX * 128 + tid * 16) = make_uint4(0, 0, 0, 0);
  else if (tid < 32) *reinterpret_cast<uint4*>(smem + PBUF_BYTES + PMAX * 128 + (tid - 16) * 16) = make_uint4(0, 0, 0, 0);

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // K range of this block: channel slices [c0, c0 + nsl)
  const int c0 = ksp * p.sl_per;
  const int nsl = (p.nsl - c0 < p.sl_per) ? p.nsl - c0 : p.sl_per;
  const int nkc = p.nsl;  // K tiles per tap in the packed weight (K order = (tap, channel))

// issue the weight tile of K tile (slice C, tap TAP) into ring slot SLOT
#define STAGE_W(C, TAP, SLOT)                                                                           \
  do {                                                                                                  \
    char* bb_ = smem + B_BASE + (SLOT) * B_BYTES + wave * 1024;                                         \
    const int woff_ = ((TAP) * nkc + (C)) * 128;                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < RB; ++i_) {                                                 \
      if (!EXACT_B || i_ < RB - 1 || grp == 0)                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (lptr_t)(bb_ + i_ * PASS_BYTES), 16, w_voff[i_], \
                                                 woff_, 0, 0);                                          \
    }                                                                                                   \
  } while (0)
// issue piece PIECE of the patch of slice C into patch buffer BUF
#define STAGE_P(C, PIECE, BUF)                                                                          \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(a_srd, (lptr_t)(smem + (BUF) * PBUF_BYTES + (PIECE) * PASS_BYTES + wave * 1024), \
                                           16, a_voff[PIECE], (C) * 128, 0, 0)

  if constexpr (LS) {
    // ---- LS schedule: all waves in lockstep, ONE barrier per K tile, no drain in steady state -----------------------
    // Ring: tile kt in slot kt % 3.  Iteration kt:  wait (this wave's loads of tile kt+1 landed; only a patch piece issued
    // in the previous iteration may stay in flight) -> barrier (everyone's share landed; everyone is done with slot
    // (kt-1) % 3) -> issue W(kt+2) into that slot (+ a piece of the next slice's patch) -> 4 k-steps of MFMAs, the
    // fragment reads running one k-step ahead ACROSS the tile boundary: the last k-step's MFMAs cover the reads of tile
    // kt+1's first fragments, so after the next barrier the matrix pipe starts at once (the de-phased schedule pays a
    // barrier + ds_read latency bubble at the head of every multiply section: profiles/r2_halo_ablation.txt).
    //   RAW W: tile kt+1 was issued in iteration kt-1 and is waited for before barrier kt, its first read follows it.
    //   RAW patch: pieces go out in taps 0..NPASS-1 <= 5, older than W(c+1, 0) (issued in tap 7) in the in-order queue.
    //   WAR: slot (kt-1) % 3 / patch buffer (c+1) & 1 were last read in iteration kt-1 / (c-1, 8), before barrier kt.
    const int nk = nsl * 9;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) STAGE_P(c0, i, 0);
    STAGE_W(c0, 0, 0);
    if (nk > 1) STAGE_W(c0, 1, 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    typename T::vec8 xf[2][MI], wf[2][NJ];
    int arow[MI], akey[MI];
    const char* bbase = smem;
#define TAP_ADDR(TAP, PBUF)                                                     \
  do {                                                                          \
    const int ky_ = (TAP) / 3, kx_ = (TAP) % 3;                                 \
    const int toff_ = (ky_ - 1) * Wo + (kx_ - 1);                               \
    const int pbase_ = (PBUF) * PBUF_BYTES;                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) {                         \
      int pp_ = ppc[i_] + toff_;                                                \
      if (kx_ == 0) pp_ = xl[i_] ? PMAX : pp_;                                  \
      if (kx_ == 2) pp_ = xr[i_] ? PMAX : pp_;                                  \
      arow[i_] = pbase_ + (pp_ << 7);                                           \
      akey[i_] = (pp_ << 3) & 0x70;                                             \
    }                                                                           \
  } while (0)
#define LOAD_FRAGS_L(KS, SET)                                                                                 \
  do {                                                                                                        \
    const int co_ = ((2 * (KS) + hi) ^ sw) * 16;                                                              \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) wf[SET][j] =                                               \
        *reinterpret_cast<const typename T::vec8*>(bbase + b_frag + j * FSTR + co_);                          \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) xf[SET][i] =                                               \
        *reinterpret_cast<const typename T::vec8*>(smem + arow[i] + ((((KS) * 32) + hik) ^ akey[i]));          \
  } while (0)
    TAP_ADDR(0, 0);
    LOAD_FRAGS_L(0, 0);
    int s_slot = 2, c_slot = 0;
    bool piece_prev = false;
    for (int cc = 0; cc < nsl; ++cc) {
      const int c = c0 + cc;
      const int pbuf = cc & 1;
      const bool next_slice = cc + 1 < nsl;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const bool more1 = next_slice || t < 8;          // a tile kt+1 exists
        const bool more2 = next_slice || t < 7;          // a tile kt+2 exists
        if (more1) {
          if (piece_prev)
            wait_vm<1>();
          else
            wait_vm<0>();
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_barrier" ::: "memory");
        }
        if (more2 && ABL != 3) {
          if (t < 7)
            STAGE_W(c, t + 2, s_slot);
          else
            STAGE_W(c + 1, t - 7, s_slot);
          s_slot = (s_slot + 1 == STAGES) ? 0 : s_slot + 1;
        }
        piece_prev = false;
        if (t < NPASS && ABL != 3) {
          if (next_slice) {
            STAGE_P(c + 1, (t < NPASS ? t : 0), pbuf ^ 1);
            piece_prev = true;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < 3) {
            if (ABL != 4) LOAD_FRAGS_L(ks + 1, (ks + 1) & 1);
          } else if (more1) {
            // fragments of the NEXT tile's first k-step (its slot / tap / patch buffer)
            c_slot = (c_slot + 1 == STAGES) ? 0 : c_slot + 1;
            bbase = smem + c_slot * B_BYTES;
            if (t < 8)
              TAP_ADDR(t + 1, pbuf);
            else
              TAP_ADDR(0, pbuf ^ 1);
            if (ABL != 4) LOAD_FRAGS_L(0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ABL == 2) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(wf[ks & 1][j]));
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(xf[ks & 1][i]));
          } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma32(wf[ks & 1][j], xf[ks & 1][i], acc[i][j]);  // D[n][m]
          }
        }
      }
    }
#undef TAP_ADDR
#undef LOAD_FRAGS_L
  } else {
  // ---- prologue: whole patch of the first slice + weight tile (c0, 0) ----
#pragma unroll
  for (int i = 0; i < NPASS; ++i) STAGE_P(c0, i, 0);
  STAGE_W(c0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // + the zero-pixel ds_writes
  asm volatile("s_barrier" ::: "memory");
  if (grp == 1) asm volatile("s_barrier" ::: "memory");  // group 1 runs one interval behind group 0

  int s_slot = 1, c_slot = 0;
  for (int cc = 0; cc < nsl; ++cc) {
    const int c = c0 + cc;
    const int pbuf = cc & 1;
    const bool next_slice = cc + 1 < nsl;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const bool more = next_slice || t < 8;
      // ---- stage: weight tile of the next K tile (+ one piece of the next slice's patch) ----
      if (more && ABL != 3) {
        if (t < 8)
          STAGE_W(c, t + 1, s_slot);
        else
          STAGE_W(c + 1, 0, s_slot);
        s_slot = (s_slot + 1 == STAGES) ? 0 : s_slot + 1;
      }
      const bool piece = (t >= 1 && t - 1 < NPASS) && next_slice && ABL != 3;
      if (t >= 1 && t - 1 < NPASS && ABL != 3) {
        if (next_slice) STAGE_P(c + 1, (t - 1 < NPASS ? t - 1 : 0), pbuf ^ 1);
      }
      if (grp == 0) {
        if (!more)
          wait_vm<0>();
        else if (piece)
          wait_vm<RB0 + 1>();
        else
          wait_vm<RB0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      // ---- multiply K tile (c, t): A fragments from the patch at the tap's offset, W fragments from the ring ----
      {
        const int ky = t / 3, kx = t % 3;  // constants after the full unroll
        const int toff = (ky - 1) * Wo + (kx - 1);
        const int pbase = pbuf * PBUF_BYTES;
        int arow[MI], akey[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          int pp = ppc[i] + toff;
          if (kx == 0) pp = xl[i] ? PMAX : pp;
          if (kx == 2) pp = xr[i] ? PMAX : pp;
          arow[i] = pbase + (pp << 7);
          akey[i] = (pp << 3) & 0x70;
        }
        const char* bbase = smem + c_slot * B_BYTES;
        c_slot = (c_slot + 1 == STAGES) ? 0 : c_slot + 1;
        typename T::vec8 xf[2][MI], wf[2][NJ];
#define LOAD_FRAGS_H(KS, SET)                                                                                 \
  do {                                                                                                        \
    const int co_ = ((2 * (KS) + hi) ^ sw) * 16;                                                              \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) wf[SET][j] =                                               \
        *reinterpret_cast<const typename T::vec8*>(bbase + b_frag + j * FSTR + co_);                          \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) xf[SET][i] =                                               \
        *reinterpret_cast<const typename T::vec8*>(smem + arow[i] + ((((KS) * 32) + hik) ^ akey[i]));          \
  } while (0)
        if (ABL != 4 || (cc == 0 && t == 0)) LOAD_FRAGS_H(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < 3 && (ABL != 4 || (cc == 0 && t == 0))) LOAD_FRAGS_H(ks + 1, (ks + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ABL == 2) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(wf[ks & 1][j]));
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(xf[ks & 1][i]));
          } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma32(wf[ks & 1][j], xf[ks & 1][i], acc[i][j]);  // D[n][m]
          }
        }
#undef LOAD_FRAGS_H
      }
      if (grp == 1) wait_vm<0>();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
    }
  }
  if (grp == 0) asm volatile("s_barrier" ::: "memory");  // pairs group 1's extra barrier
  }  // de-phased schedule
#undef STAGE_W
#undef STAGE_P

  if constexpr (ABL == 1) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
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

int ilog2_exact(int v) {  // -1 unless v is a power of two
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// geometry of a BM-row tile for this problem; false if the halo kernel cannot run it
bool halo_geometry(const dbir_gemm_desc& d, int BM, int PMAX, HParams& p) {
  if (d.mode != DBIR_MODE_CONV3X3 || d.stride != 1 || d.pad != 1 || d.upsample || d.Hi != d.Ho || d.Wi != d.Wo)
    return false;
  if (d.Cin % 64 != 0 || d.lda != d.Cin) return false;
  const int lw = ilog2_exact(d.Wo);
  if (lw < 0 || BM % d.Wo != 0) return false;
  const int R = BM / d.Wo;
  int rps, nseg;
  if (R <= d.Ho) {
    if (d.Ho % R != 0) return false;
    rps = R;
    nseg = 1;
  } else {
    if (R % d.Ho != 0) return false;
    rps = d.Ho;
    nseg = R / d.Ho;
  }
  const int lr = ilog2_exact(rps);
  if (lr < 0) return false;
  const int P = nseg * (rps + 2) * d.Wo;
  if (P > PMAX) return false;
  if ((long long)(R + 2 * nseg) * d.Wo * d.Cin * 2 >= 0x7ffffe00LL) return false;  // 32-bit patch offsets
  p.log2wo = lw;
  p.log2rps = lr;
  p.nseg = nseg;
  p.P = P;
  return true;
}

}  // namespace

// split-K second pass shared with gemm_glds.hip
int dbir_splitk_reduce_launch(const dbir_gemm_desc& d, int splitk, float* ws, hipStream_t s);

template <typename T, int WM, int WN, int MI, int NJ, int PMAX, int ABL = 0, int LS = 0>
static int launch_halo(HParams& p, hipStream_t s) {
  constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
  constexpr int ring = 2 * (PMAX * 128 + 256) + 3 * BN * 128, epi = BM * (BN + 8) * 2;
  constexpr int lds = ring > epi ? ring : epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  const dbir_gemm_desc& dd = p.d;
  if (!halo_geometry(dd, BM, PMAX, p)) {
    dbir_set_error("dbir_gemm: halo tile needs a stride-1 3x3 conv whose %d-row tiles are whole image rows "
                   "(Wo a power of two dividing %d, patch <= %d pixels)", BM, BM, PMAX);
    return DBIR_ERR_ARG;
  }
  p.nsl = dd.Cin / 64;
  int sk = dd.splitk > 1 ? dd.splitk : 1;
  if (sk > p.nsl) sk = p.nsl;
  p.sl_per = cdiv(p.nsl, sk);
  p.splitk = cdiv(p.nsl, p.sl_per);
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
  p.mtiles = cdiv(dd.M, BM);
  p.ntiles = cdiv(dd.N, BN);
  {
    // OFF by default: measured per shape in the two-stream evaluation (profiles/r4_group_m_ab.txt) GROUP_M = 4 is neutral
    // on the wide GEGLU tiles it was meant for and costs the 128x128 tile 2.4 us (6 %) on 4096 x 1280 x 1280; the L2-miss
    // traffic it removes is not what those launches wait for.  DBIR_GROUP_M=4 switches it on for experiments.
    static const int gm_env = getenv("DBIR_GROUP_M") ? atoi(getenv("DBIR_GROUP_M")) : 1;
    p.group_m = (gm_env > 1 && p.ntiles >= 8 && p.mtiles >= gm_env) ? gm_env : 1;
  }
  if (p.d.stats) {  // GroupNorm column statistics: from the epilogue (whole tiles only) or, under split-K, from the reduce pass
    if (p.splitk <= 1 && dd.M % BM == 0 && !dd.out_f32) g_dbir_stats_rows = BM;
    else if (p.splitk > 1 && dd.M % 64 == 0 && dd.batch <= 1 && !dd.out_f32) g_dbir_stats_rows = 64;
    else p.d.stats = nullptr;
  }
  auto kern = &gemm_halo_kernel<T, WM, WN, MI, NJ, PMAX, ABL, LS>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const unsigned nz = dd.batch > 0 ? dd.batch : 1;
  dim3 grid((unsigned)(p.mtiles * p.ntiles * p.splitk), nz);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, p);
  DBIR_CHECK_LAUNCH("dbir_gemm(halo)");
  if (p.splitk > 1) return dbir_splitk_reduce_launch(p.d, p.splitk, p.ws, s);
  return DBIR_OK;
}

// Can tile `tile` (50 / 51) run this descriptor?  (dbir_gemm_glds_eligible has already vetted alignment / strides.)
bool dbir_gemm_halo_eligible(const dbir_gemm_desc& d, int tile) {
  HParams p;
  if (d.act == DBIR_ACT_GEGLU && tile == 50) return false;
  return halo_geometry(d, 256, 384, p);
}

int dbir_gemm_halo(const dbir_gemm_desc& dd, int tile, hipStream_t s) {
  HParams p;
  p.d = dd;
  p.vec_bias = dd.bias && (reinterpret_cast<uintptr_t>(dd.bias) & 15) == 0;
  p.vec_rv = dd.rowvec && (reinterpret_cast<uintptr_t>(dd.rowvec) & 7) == 0 && dd.rowvec_ld % 4 == 0;
  p.ws = reinterpret_cast<float*>(dd.ws);
  p.a_elems = (long long)dd.B * dd.Hi * dd.Wi * dd.Cin;
  const bool f16 = dd.dtype == DBIR_F16;
  switch (tile) {
    case 50:  // 256x160, 8 waves (32x160 each): N = 320 k without padding
      return f16 ? launch_halo<F16, 8, 1, 1, 5, 384>(p, s) : launch_halo<BF16, 8, 1, 1, 5, 384>(p, s);
    case 51:  // 256x128, 8 waves (64x64 each)
      return f16 ? launch_halo<F16, 4, 2, 2, 2, 384>(p, s) : launch_halo<BF16, 4, 2, 2, 2, 384>(p, s);
    case 54:  // 256x32, 8 waves (32x32 each): heads with a handful of output channels (UNet `out`: 320 -> 4, f32 store)
      return f16 ? launch_halo<F16, 8, 1, 1, 1, 384, 0, 1>(p, s) : launch_halo<BF16, 8, 1, 1, 1, 384, 0, 1>(p, s);
    case 52:  // 256x160, lockstep schedule with cross-tile fragment prefetch
      return f16 ? launch_halo<F16, 8, 1, 1, 5, 384, 0, 1>(p, s) : launch_halo<BF16, 8, 1, 1, 5, 384, 0, 1>(p, s);
    case 53:  // 256x128, lockstep schedule
      return f16 ? launch_halo<F16, 4, 2, 2, 2, 384, 0, 1>(p, s) : launch_halo<BF16, 4, 2, 2, 2, 384, 0, 1>(p, s);
#ifdef DBIR_DIAG  // diagnostic ablations (f16 only, outputs are meaningless): only in a `DBIR_DIAG=1 sh build.sh` library,
                  // never reachable through the production ABI (tools/probes/halo_ablate.py)
    case 64: return launch_halo<F16, 8, 1, 1, 5, 384, 1, 1>(p, s);   // ablations of tile 52
    case 65: return launch_halo<F16, 8, 1, 1, 5, 384, 2, 1>(p, s);
    case 66: return launch_halo<F16, 8, 1, 1, 5, 384, 3, 1>(p, s);
    case 67: return launch_halo<F16, 8, 1, 1, 5, 384, 4, 1>(p, s);
    // diagnostic ablations of tile 50 (f16 only; outputs are meaningless): tools/probes/halo_ablate.py
    case 60: return launch_halo<F16, 8, 1, 1, 5, 384, 1>(p, s);
    case 61: return launch_halo<F16, 8, 1, 1, 5, 384, 2>(p, s);
    case 62: return launch_halo<F16, 8, 1, 1, 5, 384, 3>(p, s);
    case 63: return launch_halo<F16, 8, 1, 1, 5, 384, 4>(p, s);
#endif
  }
  dbir_set_error("dbir_gemm: bad halo tile %d", tile);
  return DBIR_ERR_ARG;
}
