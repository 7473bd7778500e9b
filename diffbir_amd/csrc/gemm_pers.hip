// Persistent linear (dense A [M, K] x packed W [N, K]) kernel for the small-K projections of the transformer blocks.
//
// Why a second kernel for the same math: with K = 320 .. 1280 an output tile has 5 .. 20 K tiles, and the
// one-tile-per-workgroup kernels (gemm_glds.hip) pay, per tile, the HBM latency of the prologue, a main loop whose
// 2-slot ring keeps one K tile in flight (latency-bound: ~2.5 us per K tile instead of ~1), and an epilogue (LDS
// transpose + stores) during which nothing is in flight — 16 us of fixed cost per tile, measured.  The
// vendor library lands on the same numbers (tools/blas_yardstick.py): 36 us for 65536x320x320 = 84 MB of traffic,
// 2.3 TB/s, while a streaming kernel moves those bytes in 19.5 us.
//
// This kernel keeps the memory pipe busy across tile boundaries:
//   * persistent workgroups: a workgroup walks its list of output tiles; the K tiles of ALL its output tiles form one
//     flat stream through a STAGES-slot LDS ring (K depth 32, direct-to-LDS `buffer_load ... lds`, swizzle on the
//     source side as in gemm_glds.hip) with STAGES-1 K tiles always in flight — the first K tiles of output tile i+1 are
//     fetched under the last MFMAs and the whole epilogue of tile i;
//   * the epilogue never touches LDS and has no barrier: accumulators are D[n][m] (a lane holds 4 consecutive output
//     columns of a row, the two half-waves interleave 4-column blocks); `v_permlane32_swap` exchanges the packed 16-bit
//     pairs between the half-waves so that every lane owns 8 consecutive columns = one 16-byte store (and one 16-byte
//     residual load), issued straight from registers.  Same rounding points as the LDS-transposed epilogue (round to
//     16 bit, add the residual in f32, round again): results are bit-identical to the other tiles;
//   * vmcnt counts loads AND stores on gfx9, loads return in order: before its stores a wave drains its loads
//     (`vmcnt(0)`: the prefetched K tiles have had the whole epilogue arithmetic to land), so the next output tile's
//     first STAGES-1 K tiles need no wait, and the counted waits that follow only ever see loads issued after the
//     stores (outstanding stores can only make a counted wait conservative, never wrong);
//   * 4-wave variants fit two workgroups per CU (<= 80 KB LDS): one workgroup's epilogue arithmetic (GEGLU: an erf per
//     output) runs under the other's MFMAs.
//   * tile order is XCD-aware: an XCD owns a contiguous range of tiles, neighbouring workgroups take the column tiles
//     of the same activation panel at the same time (one HBM fetch of the panel per XCD L2).
// Eligibility (dbir_gemm_pers_eligible): linear, K % 32 == 0, M a multiple of the tile height, N % 8 == 0, 16-byte
// aligned row-major 16-bit output / residual, no row vector, no split-K, no transposed or f32 store.
#include <stdint.h>
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

struct PParams {
  dbir_gemm_desc d;
  int mtiles, ntiles, nk;  // output tiles, K tiles per output tile
  int q, gx;               // tiles / workgroups per XCD
  int a_bytes, w_bytes;    // buffer descriptor extents
};

typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void pwait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int WM, int WN, int MI, int NJ, int STAGES, int OCC>
__global__ __launch_bounds__(64 * WM * WN, OCC) void gemm_pers_kernel(const PParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
  constexpr int BK = 32, ROWB = 64, CPR = 4, KS = 2;
  constexpr int RPP = NT / CPR;  // tile rows covered by one pass of the whole workgroup
  constexpr int PASS_BYTES = NT * 16;
  constexpr int RA = BM / RPP, RB = (BN + RPP - 1) / RPP, LOADS = RA + RB;
  constexpr int A_BYTES = BM * ROWB, BUF_BYTES = A_BYTES + RB * RPP * ROWB;
  constexpr int LA = STAGES - 1;  // K tiles in flight
  static_assert(BM % RPP == 0, "activation tile rows must be a multiple of the pass height");
  static_assert(LA >= 1 && LA <= 5 && LA * LOADS < 64, "vmcnt is 6 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const dbir_gemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lq = lane & 31, hi = lane >> 5;

  // ---- this workgroup's output tiles: XCD x owns tiles [x*q, (x+1)*q); workgroup `loc` of the XCD takes loc, loc+gx, ..
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int x0 = xcd * p.q;
  int xn = p.mtiles * p.ntiles - x0;
  xn = xn > p.q ? p.q : xn;
  const int nmine = xn > loc ? (xn - loc + p.gx - 1) / p.gx : 0;
  if (nmine == 0) return;
  const int nk = p.nk, total = nmine * nk;

  constexpr int OOB = 0x7fffff00;
  const __amdgpu_buffer_rsrc_t a_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.A), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.W), 0, p.w_bytes, 0x00020000);

  // staging role of a thread: LDS chunk position (row = tid / 4 + RPP * i, cpos = tid & 3), logical chunk cpos ^ key
  const int srow = tid / CPR;
  const int cch = ((tid % CPR) ^ ((srow >> 2) & 3)) * 8;
  int a_voff[RA], w_voff[RB];
#pragma unroll
  for (int i = 0; i < RA; ++i) a_voff[i] = (int)(((long long)(srow + RPP * i) * d.lda + cch) * 2);
#pragma unroll
  for (int i = 0; i < RB; ++i) w_voff[i] = ((srow + RPP * i) * d.Kpad + cch) * 2;

  // staging cursor (uniform): output tile ordinal, K tile, ring slot, byte offsets of the tile's operand panels
  int s_o = 0, s_kt = 0, s_slot = 0, s_abase, s_wbase, s_wlim, issued = 0;  // s_wlim: valid weight rows of the tile
  {
    const int lid = x0 + loc, tm = lid / p.ntiles, tn = lid - tm * p.ntiles;
    s_abase = (int)((long long)tm * BM * d.lda * 2);
    s_wbase = tn * BN * d.Kpad * 2;
    s_wlim = d.Wrows - tn * BN < BN ? d.Wrows - tn * BN : BN;
  }
#define PSTAGE()                                                                                            \
  do {                                                                                                      \
    char* ab_ = smem + s_slot * BUF_BYTES + wave * 1024;                                                    \
    char* bb_ = ab_ + A_BYTES;                                                                              \
    const int ka_ = s_abase + s_kt * (BK * 2), kw_ = s_wbase + s_kt * (BK * 2);                             \
    _Pragma("unroll") for (int i = 0; i < RA; ++i)                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_srd, (lptr_t)(ab_ + i * PASS_BYTES), 16, a_voff[i], ka_, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (lptr_t)(bb_ + i * PASS_BYTES), 16,                  \
                                                 srow + RPP * i < s_wlim ? w_voff[i] : OOB, kw_, 0, 0);     \
    ++issued;                                                                                               \
    s_slot = (s_slot + 1 == STAGES) ? 0 : s_slot + 1;                                                       \
    if (++s_kt == nk) {                                                                                     \
      s_kt = 0;                                                                                             \
      ++s_o;                                                                                                \
      const int lid_ = x0 + loc + s_o * p.gx, tm_ = lid_ / p.ntiles, tn_ = lid_ - tm_ * p.ntiles;           \
      s_abase = (int)((long long)tm_ * BM * d.lda * 2);                                                     \
      s_wbase = tn_ * BN * d.Kpad * 2;                                                                      \
      s_wlim = d.Wrows - tn_ * BN < BN ? d.Wrows - tn_ * BN : BN;                                           \
    }                                                                                                       \
  } while (0)

  // fragment read offsets (bytes) inside a slot: row * 64 + ((2 * ks + hi) ^ key(row)) * 16
  const int a_frag = (wm * 32 * MI + lq) * ROWB;
  const int b_frag = A_BYTES + (wn * 32 * NJ + lq) * ROWB;
  const int sw = (lq >> 2) & 3;
  constexpr int FSTR = 32 * ROWB;

  const u16* __restrict__ Rg = reinterpret_cast<const u16*>(d.R);
  u16* __restrict__ Cg = reinterpret_cast<u16*>(d.C);
  const bool geglu = d.act == DBIR_ACT_GEGLU;
  const int n_out = geglu ? d.N / 2 : d.N;

  // The bias is the INITIAL VALUE of the accumulators (f32): its loads are issued where their latency is free (the
  // prologue; for later tiles the epilogue of the previous tile, straight into the dead accumulator registers) instead
  // of serialising the epilogue.  Columns past N (last column tile) read a clamped address and are never stored.
  f32x16 acc[MI][NJ];
#define PLOAD_BIAS(TN)                                                                                        \
  do {                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) _Pragma("unroll") for (int g = 0; g < 4; ++g) {            \
      int n0_ = (TN) * BN + wn * 32 * NJ + j * 32 + 8 * g + 4 * hi;                                           \
      n0_ = n0_ < d.N - 4 ? n0_ : d.N - 4;                                                                    \
      const float4 b_ = *reinterpret_cast<const float4*>(biasp + n0_);                                        \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                        \
        acc[i][j][4 * g + 0] = b_.x;                                                                          \
        acc[i][j][4 * g + 1] = b_.y;                                                                          \
        acc[i][j][4 * g + 2] = b_.z;                                                                          \
        acc[i][j][4 * g + 3] = b_.w;                                                                          \
      }                                                                                                       \
    }                                                                                                         \
  } while (0)
#define PZERO_ACC()                                                                                           \
  do {                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)             \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;                                    \
  } while (0)
// Tell the compiler's waitcnt pass that the accumulators are consumed HERE (it cannot see through the explicit
// s_waitcnt): otherwise it protects their first use in the K loop with a vmcnt(0) of its own, which would also wait for
// the previous tile's stores.
#define PTOUCH_ACC()                                                                                          \
  do {                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)             \
        asm volatile("" : "+v"(acc[i][j]));                                                                   \
  } while (0)
  const float* __restrict__ biasp = d.bias;
  if (biasp) {
    const int lid = x0 + loc;
    PLOAD_BIAS(lid - (lid / p.ntiles) * p.ntiles);
  } else {
    PZERO_ACC();
  }

#pragma unroll
  for (int s = 0; s < LA; ++s)
    if (issued < total) PSTAGE();
  PTOUCH_ACC();

  int consumed = 0, landed = 0, c_slot = 0;
  for (int o = 0; o < nmine; ++o) {
    for (int kt = 0; kt < nk; ++kt) {
      if (consumed >= landed) {  // not covered by the drain of the previous epilogue: counted wait (loads return in order)
        const int after = issued - consumed - 1;  // K tiles issued after the one consumed now
        if (after >= 4) pwait_vmcnt<(LA >= 5 ? 4 : 0) * LOADS>();
        else if (after == 3) pwait_vmcnt<(LA >= 4 ? 3 : 0) * LOADS>();
        else if (after == 2) pwait_vmcnt<(LA >= 3 ? 2 : 0) * LOADS>();
        else if (after == 1) pwait_vmcnt<(LA >= 2 ? 1 : 0) * LOADS>();
        else pwait_vmcnt<0>();
      }
      // every wave's share of this K tile landed, and everyone is done reading the slot refilled next
      asm volatile("s_barrier" ::: "memory");
      const char* base = smem + c_slot * BUF_BYTES;
      c_slot = (c_slot + 1 == STAGES) ? 0 : c_slot + 1;
      typename T::vec8 xf[2][MI], wf[2][NJ];
#define PLOAD_FRAGS(KSI, SET)                                                                                \
  do {                                                                                                       \
    const int co_ = ((2 * (KSI) + hi) ^ sw) * 16;                                                            \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) wf[SET][j] =                                              \
        *reinterpret_cast<const typename T::vec8*>(base + b_frag + j * FSTR + co_);                          \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) xf[SET][i] =                                              \
        *reinterpret_cast<const typename T::vec8*>(base + a_frag + i * FSTR + co_);                          \
  } while (0)
      PLOAD_FRAGS(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (issued < total) PSTAGE();  // refills the slot read one iteration ago (ordered by the barrier above)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks < KS - 1) PLOAD_FRAGS(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma32(wf[ks & 1][j], xf[ks & 1][i], acc[i][j]);  // D[n][m]
      }
#undef PLOAD_FRAGS
      ++consumed;
    }

    // ---------------- epilogue of output tile o, straight from registers ----------------
    const int lid = x0 + loc + o * p.gx, tm = lid / p.ntiles, tn = lid - tm * p.ntiles;
    // after the half-wave exchange a lane owns columns [8 * (2 * gp + hi), +8) of 32-column block j, gp = 0, 1
    constexpr int NJE = 2;  // GEGLU: (value, gate) column blocks, NJ == 2 only
    uint4 outv[MI][NJ][2];
    uint4 resv[MI][NJ][2];
    if (Rg) {  // residual loads first (clamped columns: branch-free); they land under the arithmetic below
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const long long m = (long long)tm * BM + wm * 32 * MI + i * 32 + lq;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (NJ == NJE && geglu && (j & 1)) continue;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            int col = geglu ? tn * (BN / 2) + wn * 16 * NJ + 8 * (2 * gp + hi)
                            : tn * BN + wn * 32 * NJ + j * 32 + 8 * (2 * gp + hi);
            col = col < n_out - 8 ? col : n_out - 8;
            resv[i][j][gp] = *reinterpret_cast<const uint4*>(Rg + m * d.ldr + col);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (NJ == NJE && geglu && (j & 1)) continue;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        uint2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
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
            if constexpr (NJ == NJE) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= gelu_fast(acc[i][1][4 * g + e]);
            }
          }
          pk[g].x = T::pack2(v[0] * d.out_scale, v[1] * d.out_scale);
          pk[g].y = T::pack2(v[2] * d.out_scale, v[3] * d.out_scale);
        }
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          // X = block g = 2gp (low lanes: cols 8g..8g+3, high lanes: 8g+4..8g+7), Y = block g+1: swap X.high <-> Y.low
          const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * gp].x, pk[2 * gp + 1].x, false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * gp].y, pk[2 * gp + 1].y, false, false);
          outv[i][j][gp] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      }
    }
    if (biasp && o + 1 < nmine) {  // the accumulators are dead: fetch the next tile's bias into them
      const int lidn = lid + p.gx;
      PLOAD_BIAS(lidn - (lidn / p.ntiles) * p.ntiles);
    } else {
      PZERO_ACC();
    }
    // loads return in order: once nothing is outstanding, the residual, the bias AND every K tile prefetched so far
    // have landed (they have had the whole arithmetic above to do so)
    pwait_vmcnt<0>();
    PTOUCH_ACC();
    landed = issued;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const long long m = (long long)tm * BM + wm * 32 * MI + i * 32 + lq;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (NJ == NJE && geglu && (j & 1)) continue;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int col = geglu ? tn * (BN / 2) + wn * 16 * NJ + 8 * (2 * gp + hi)
                                : tn * BN + wn * 32 * NJ + j * 32 + 8 * (2 * gp + hi);
          uint4 v = outv[i][j][gp];
          if (Rg) {
            float a[8], b[8];
            unpack8<T>(v, a);
            unpack8<T>(resv[i][j][gp], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];
            v = pack8<T>(a);
          }
          if (col < n_out) *reinterpret_cast<uint4*>(Cg + m * d.ldc + col) = v;
        }
      }
    }
  }
#undef PLOAD_BIAS
#undef PZERO_ACC
#undef PTOUCH_ACC
#undef PSTAGE
#endif  // __HIP_DEVICE_COMPILE__
}

template <int WM, int WN, int MI, int NJ>
bool pers_shape_ok(const dbir_gemm_desc& d) {
  constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
  if (d.mode != DBIR_MODE_LINEAR || d.batch > 1 || d.splitk > 1 || d.out_f32 || d.store_mode != 0 || d.rowvec) return false;
  if (d.K % 32 != 0 || d.Kpad < d.K || d.M % BM != 0 || d.N % 8 != 0 || d.ldc % 8 != 0 || d.lda % 8 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.W) | reinterpret_cast<uintptr_t>(d.C)) & 15) return false;
  if (d.R && ((reinterpret_cast<uintptr_t>(d.R) & 15) || d.ldr % 8 != 0)) return false;
  if (d.bias && (reinterpret_cast<uintptr_t>(d.bias) & 15)) return false;
  if (d.act == DBIR_ACT_GEGLU && (NJ != 2 || d.N % 64 != 0)) return false;
  if (((long long)(d.M - 1) * d.lda + d.K) * 2 >= 0x7ffffe00LL || (long long)d.Wrows * d.Kpad * 2 >= 0x7ffffe00LL) return false;
  return true;
}

template <typename T, int WM, int WN, int MI, int NJ, int STAGES, int OCC>
int launch_pers(const dbir_gemm_desc& dd, hipStream_t s) {
  constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN, NT = 64 * WM * WN;
  constexpr int RPP = NT / 4, RB = (BN + RPP - 1) / RPP;
  constexpr int lds = STAGES * (BM * 64 + RB * RPP * 64);
  static_assert(lds * OCC <= 160 * 1024, "LDS budget");
  if (!pers_shape_ok<WM, WN, MI, NJ>(dd)) {
    dbir_set_error("dbir_gemm: persistent linear tile needs a dense linear with K %% 32 == 0, M %% %d == 0, N %% 8 == 0, "
                   "16-byte aligned 16-bit row-major output / residual, no row vector / split-K / transposed store", BM);
    return DBIR_ERR_ARG;
  }
  PParams p;
  p.d = dd;
  p.mtiles = dd.M / BM;
  p.ntiles = cdiv(dd.N, BN);
  p.nk = dd.K / 32;
  const int tiles = p.mtiles * p.ntiles;
  p.q = cdiv(tiles, 8);
  p.gx = p.q < 32 * OCC ? p.q : 32 * OCC;
  p.a_bytes = (int)((((long long)(dd.M - 1) * dd.lda + dd.K) * 2 + 15) & ~15LL);
  p.w_bytes = (int)((long long)dd.Wrows * dd.Kpad * 2);
  auto kern = &gemm_pers_kernel<T, WM, WN, MI, NJ, STAGES, OCC>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * p.gx)), dim3(NT), lds, s, p);
  DBIR_CHECK_LAUNCH("dbir_gemm(persistent)");
  return DBIR_OK;
}

template <typename T>
int dispatch_pers(const dbir_gemm_desc& d, int tile, hipStream_t s) {
  switch (tile) {
    case 70: return launch_pers<T, 8, 1, 1, 5, 4, 1>(d, s);  // 256x160, 8 waves, 4-slot ring (128 KB)
    case 71: return launch_pers<T, 4, 1, 1, 5, 4, 2>(d, s);  // 128x160, 4 waves, 4-slot ring (80 KB): 2 workgroups / CU
    case 72: return launch_pers<T, 4, 2, 2, 2, 5, 1>(d, s);  // 256x128, 8 waves, 5-slot ring (120 KB)
    case 73: return launch_pers<T, 2, 2, 2, 2, 5, 2>(d, s);  // 128x128, 4 waves, 5-slot ring (80 KB): 2 workgroups / CU
  }
  dbir_set_error("dbir_gemm: bad persistent tile %d", tile);
  return DBIR_ERR_ARG;
}

}  // namespace

bool dbir_gemm_pers_eligible(const dbir_gemm_desc& d, int tile) {
  switch (tile) {
    case 70: return pers_shape_ok<8, 1, 1, 5>(d);
    case 71: return pers_shape_ok<4, 1, 1, 5>(d);
    case 72: return pers_shape_ok<4, 2, 2, 2>(d);
    case 73: return pers_shape_ok<2, 2, 2, 2>(d);
  }
  return false;
}

int dbir_gemm_pers(const dbir_gemm_desc& d, int tile, hipStream_t s) {
  return d.dtype == DBIR_F16 ? dispatch_pers<F16>(d, tile, s) : dispatch_pers<BF16>(d, tile, s);
}
