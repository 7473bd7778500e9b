// Phased implicit GEMM (linear / 1x1 / 3x3 convolution) for gfx950: 256x256x64 block tile, 8 wave64 in two
// STAGGERED groups, direct-to-LDS staging with counted vmcnt (cdna_hip_programming.md §5 "8-phase" structure,
// T2 + T3/T4 + T5), re-derived for this engine's operand layout.  Selected with tile id 13.
//
// Why: in gemm_glds.hip all waves of a block run in lockstep — every K tile each wave issues its loads, waits for
// its ds_reads and only then feeds the matrix pipe, so the two waves that share a SIMD stall together.  Here a K
// tile is cut into 4 phases (one 64x32 quadrant of the wave's 128x64 output each: 8 v_mfma_f32_32x32x16) and the
// block's waves form two groups that run ONE BARRIER apart: while group 0 multiplies phase p, group 1 issues the
// ds_reads + the direct-to-LDS loads of its phase p, and vice versa — each SIMD hosts one wave of each group, so
// its matrix pipe alternates between them instead of idling, and s_setprio(1) around the MFMA cluster lets the
// multiplying wave win issue arbitration.
//
// LDS (128 KB): 2 K-tile buffers x { A-lo, A-hi, B-lo, B-hi } half tiles of 128 rows x 128 B (64 halfs), rows
// XOR-swizzled in 16-byte chunks with key (row >> 1) & 7 on the SOURCE side (glds writes lane-linear) and on the
// ds_read_b128 (conflict-free; same image as gemm_glds.hip).
//   wave (grp, wc): output rows  mh*128 + grp*64 + [0,64)   (mh = 0,1  -> A-lo / A-hi)
//                   output cols  wc*64 + nh*32 + [0,32)     (nh = 0,1  -> B-lo / B-hi; B-half row rho holds logical
//                   tile column ((rho>>5)&3)*64 + (rho>>7)*32 + (rho&31), so a wave's value/gate GEGLU pair is local)
//   phase q of K tile t (buffer b = t&1):  reads (then MFMA quadrant)          stages (2 glds per thread)
//     q0: B-lo(4) + A-lo(8) ds_read_b128   (mh0,nh0)                            B-hi(t+1) -> buffer b^1
//     q1: B-hi(4)                          (mh0,nh1)                            A-hi(t+1) -> buffer b^1
//     q2: A-hi(8)                          (mh1,nh1)                            A-lo(t+2) -> buffer b
//     q3: -  (B-lo fragments kept)         (mh1,nh0)                            B-lo(t+2) -> buffer b
// Hazards (I_k = interval between workgroup barriers k and k+1; group 0 reads phase p in I_2p and multiplies in
// I_2p+1, group 1 one interval later):
//   WAR  a half tile is restaged >= 2 phases after the phase that last read it (reads are retired by
//        lgkmcnt(0) right after the reading phase's first barrier, one interval before the earliest restage);
//   RAW  every wave waits `vmcnt(8)` (the 4 most recent half tiles may stay in flight) BEFORE the first barrier of
//        the phase preceding the first read of the half tile that the wait retires; loads past the last K tile are
//        issued from a zero page into (free) slots so the count is the same in every phase.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BK = 64;
#ifdef DBIR_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif
constexpr int PBM = 256, PBN = 256, PNT = 512;
constexpr int HALF_BYTES = 128 * 128;        // 128 rows x 128 B
constexpr int OP_BYTES = 2 * HALF_BYTES;     // one operand of one K tile
constexpr int BUF_BYTES = 2 * OP_BYTES;      // one K tile
constexpr int EPI_BYTES = PBM * (PBN + 8) * 2;
constexpr int LDS_BYTES = (2 * BUF_BYTES > EPI_BYTES) ? 2 * BUF_BYTES : EPI_BYTES;

__device__ __attribute__((aligned(256))) unsigned int g_zero_page_ph[64];

struct PhParams {
  dbir_gemm_desc d;
  int Hv, Wv;
  int nkc, ntaps;
  int mtiles, ntiles;
  int vec_bias, vec_rv;
  int tap_inner;
  int debug;  // DIAGNOSTIC (env DBIR_GEMM_DEBUG=5): per-wave s_memtime phase accumulators into d.ws
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float gelu_fast_ph(float x) {  // same polynomial erf as gemm_glds.hip
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float e = 1.0f - poly * t * __expf(-z * z);
  const float erfv = x < 0.f ? -e : e;
  return 0.5f * x * (1.0f + erfv);
}

#define PH_BARRIER() asm volatile("s_barrier" ::: "memory")
#define PH_WAIT_VM8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
// end of a phase's load part -> multiply part
#define PH_TS(ACC)                                                \
  do {                                                            \
    if (instr) {                                                  \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      ACC += now_ - tprev;                                        \
      tprev = now_;                                               \
    }                                                             \
  } while (0)
#define PH_ENTER_MMA()                               \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    PH_TS(tacc0);                                    \
    PH_BARRIER();                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    PH_TS(tacc1);                                    \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_setprio(1);                   \
  } while (0)
#define PH_LEAVE_MMA()                 \
  do {                                 \
    __builtin_amdgcn_s_setprio(0);     \
    __builtin_amdgcn_sched_barrier(0); \
    PH_TS(tacc2);                      \
    PH_BARRIER();                      \
    PH_TS(tacc3);                      \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)

template <typename T>
__global__ __launch_bounds__(PNT) void gemm_ph_kernel(const PhParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename T::vec8 vec8;
  const dbir_gemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int lq = lane & 31, hi = lane >> 5;
  const int bz = blockIdx.y;

  int tm, tn;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nwg >> 3, r = nwg & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    tn = lid % p.ntiles;
    tm = lid / p.ntiles;
  }
  const int M = d.M;
  const u16* __restrict__ Ag = reinterpret_cast<const u16*>(d.A) + (long long)bz * d.strideA_z;
  const u16* __restrict__ Wg = reinterpret_cast<const u16*>(d.W) + (long long)bz * d.strideW_z;
  const u16* zp = reinterpret_cast<const u16*>(g_zero_page_ph);
  const bool conv = d.mode == DBIR_MODE_CONV3X3;
  const int nk = p.nkc * p.ntaps;

  // ---- staging roles: LDS row rho = srow + 64*j (j = 0..3; j>>1 = half), chunk position tid&7 ----
  const int srow = tid >> 3;
  const int cch = ((tid & 7) ^ ((srow >> 1) & 7)) * 8;  // logical K offset (halfs) fetched by this thread
  // activation rows: pointer to the element of tap (0,0) + 9-bit tap validity mask (+ x/y parity bits 16/17 for the
  // nearest-x2 upsampled input) — same scheme as gemm_glds.hip
  const u16* a_base[4];
  unsigned a_mask[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = tm * PBM + srow + 64 * j;
    const bool ok = m < M;
    if (conv) {
      const int hw = d.Ho * d.Wo;
      const int mm = ok ? m : 0;
      const int b = mm / hw;
      const int rem = mm - b * hw;
      const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
      const int iy0 = oy * d.stride - d.pad, ix0 = ox * d.stride - d.pad;
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = iy0 + t / 3, ix = ix0 + t % 3;
        if (ok && iy >= 0 && iy < p.Hv && ix >= 0 && ix < p.Wv) mk |= 1u << t;
      }
      int sy = iy0, sx = ix0;
      if (d.upsample) {
        mk |= (unsigned)(ix0 & 1) << 16 | (unsigned)(iy0 & 1) << 17;
        sy >>= 1;
        sx >>= 1;
      }
      a_mask[j] = mk;
      a_base[j] = Ag + ((long long)b * d.Hi * d.Wi + (long long)sy * d.Wi + sx) * d.Cin + cch;
    } else {
      a_mask[j] = ok ? 1u : 0u;
      a_base[j] = Ag + (long long)(ok ? m : 0) * d.lda + cch;
    }
  }
  const u16* w_rp[4];
  bool w_rv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rho = srow + 64 * j;
    const int n = tn * PBN + ((rho >> 5) & 3) * 64 + (rho >> 7) * 32 + (rho & 31);
    w_rv[j] = n < d.Wrows;
    w_rp[j] = w_rv[j] ? Wg + (long long)n * d.Kpad + cch : zp;
  }

  // per-half-tile cursors (K tile, tap, channel tile); K tiles are visited tap-inner for convolutions (p.tap_inner,
  // see gemm_glds.hip).  Index 0/1 = activation lo/hi, 2/3 = weight lo/hi.
  int c_kt[4] = {0, 0, 0, 0}, c_tap[4] = {0, 0, 0, 0}, c_cc[4] = {0, 0, 0, 0};

#define PH_ADVANCE(C)                    \
  do {                                   \
    ++c_kt[C];                           \
    if (p.tap_inner) {                   \
      if (++c_tap[C] == p.ntaps) {       \
        c_tap[C] = 0;                    \
        ++c_cc[C];                       \
      }                                  \
    } else if (++c_cc[C] == p.nkc) {     \
      c_cc[C] = 0;                       \
      ++c_tap[C];                        \
    }                                    \
  } while (0)

// stage the next K tile of activation half H into buffer BUFI (0/1), then advance that half's cursor
#define PH_STAGE_A(H, BUFI)                                                                       \
  do {                                                                                            \
    char* dst_ = smem + (BUFI) * BUF_BYTES + (H) * HALF_BYTES + wave * 1024;                      \
    const bool live_ = c_kt[H] < nk;                                                              \
    const int tap_ = c_tap[H];                                                                    \
    const int ky_ = (tap_ * 11) >> 5, kx_ = tap_ - 3 * ky_;                                       \
    const long long aoff_ = (long long)(ky_ * d.Wi + kx_) * d.Cin + c_cc[H] * BK;                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                            \
      const int j_ = 2 * (H) + i_;                                                                \
      long long o_ = aoff_;                                                                       \
      if (d.upsample) {                                                                           \
        const int dy_ = (ky_ + (int)((a_mask[j_] >> 17) & 1)) >> 1;                               \
        const int dx_ = (kx_ + (int)((a_mask[j_] >> 16) & 1)) >> 1;                               \
        o_ = (long long)(dy_ * d.Wi + dx_) * d.Cin + c_cc[H] * BK;                                \
      }                                                                                           \
      const bool v_ = live_ && ((a_mask[j_] >> tap_) & 1u);                                       \
      const u16* src_ = v_ ? a_base[j_] + o_ : zp;                                                \
      __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(dst_ + i_ * 8192), 16, 0, 0);       \
    }                                                                                             \
    if (live_) PH_ADVANCE(H);                                                                     \
  } while (0)

// stage the next K tile of weight half H into buffer BUFI
#define PH_STAGE_B(H, BUFI)                                                                       \
  do {                                                                                            \
    char* dst_ = smem + (BUFI) * BUF_BYTES + OP_BYTES + (H) * HALF_BYTES + wave * 1024;           \
    const bool live_ = c_kt[2 + (H)] < nk;                                                        \
    const long long woff_ = (long long)(c_tap[2 + (H)] * p.nkc + c_cc[2 + (H)]) * BK;             \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                            \
      const int j_ = 2 * (H) + i_;                                                                \
      const u16* src_ = (live_ && w_rv[j_]) ? w_rp[j_] + woff_ : zp;                              \
      __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(dst_ + i_ * 8192), 16, 0, 0);       \
    }                                                                                             \
    if (live_) PH_ADVANCE(2 + (H));                                                               \
  } while (0)

  f32x16 acc[2][2][2];  // [mh][nh][i]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][i][r] = 0.f;

  // fragment read offsets (bytes): row * 128 + ((2*ks + hi) ^ key(row)) * 16, key = (lq >> 1) & 7 for every block row
  const int cbase = ((hi ^ ((lq >> 1) & 7)) & 7) * 16;
  const int a_frag = (grp * 64 + lq) * 128;
  const int b_frag = OP_BYTES + (wc * 32 + lq) * 128;

  vec8 af[2][4], bl[4], bh[4];
#define PH_READ_A(MH, BASE)                                                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)     \
      af[i_][ks_] = *reinterpret_cast<const vec8*>((BASE) + a_frag + (MH) * HALF_BYTES + i_ * 4096 +     \
                                                   (cbase ^ (ks_ * 32)))
#define PH_READ_B(DST, NH, BASE)                                                                         \
  _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                    \
      DST[ks_] = *reinterpret_cast<const vec8*>((BASE) + b_frag + (NH) * HALF_BYTES + (cbase ^ (ks_ * 32)))
#define PH_MMA(MH, NH, BF)                                                                               \
  _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)     \
      acc[MH][NH][i_] = T::mfma32(BF[ks_], af[i_][ks_], acc[MH][NH][i_])  /* D[n][m] */

  // ---- prologue: K tile 0 complete + the lo halves of K tile 1 (issue order matters for the counted waits) ----
  PH_STAGE_A(0, 0);
  PH_STAGE_B(0, 0);
  PH_STAGE_B(1, 0);
  PH_STAGE_A(1, 0);
  PH_STAGE_A(0, 1);
  PH_STAGE_B(0, 1);
  PH_WAIT_VM8();  // A-lo(0), B-lo(0) of this wave landed
  PH_BARRIER();
  if (grp == 1) PH_BARRIER();  // group 1 runs one barrier behind group 0

  unsigned long long tacc0 = 0, tacc1 = 0, tacc2 = 0, tacc3 = 0, tprev = 0;
  const bool instr = kDiag && p.debug == 5;
  if (instr) tprev = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = tprev;
  for (int t = 0; t < nk; ++t) {
    const int b = t & 1;
    const char* base = smem + b * BUF_BYTES;
    // ---- q0 ----
    PH_READ_B(bl, 0, base);
    PH_READ_A(0, base);
    PH_STAGE_B(1, b ^ 1);
    PH_WAIT_VM8();  // B-hi(t)
    PH_ENTER_MMA();
    PH_MMA(0, 0, bl);
    PH_LEAVE_MMA();
    // ---- q1 ----
    PH_READ_B(bh, 1, base);
    PH_STAGE_A(1, b ^ 1);
    PH_WAIT_VM8();  // A-hi(t)
    PH_ENTER_MMA();
    PH_MMA(0, 1, bh);
    PH_LEAVE_MMA();
    // ---- q2 ----
    PH_READ_A(1, base);
    PH_STAGE_A(0, b);
    PH_ENTER_MMA();
    PH_MMA(1, 1, bh);
    PH_LEAVE_MMA();
    // ---- q3 ----
    PH_STAGE_B(0, b);
    PH_WAIT_VM8();  // A-lo(t+1), B-lo(t+1)
    PH_ENTER_MMA();
    PH_MMA(1, 0, bl);
    PH_LEAVE_MMA();
  }
  if (instr && d.ws && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(d.ws) + ((long long)blockIdx.x * 8 + wave) * 8;
    o[0] = tacc0; o[1] = tacc1; o[2] = tacc2; o[3] = tacc3;
    o[4] = __builtin_amdgcn_s_memtime() - tstart; o[5] = nk;
  }
  if (grp == 0) PH_BARRIER();  // pair group 1's extra barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // zero-page tail loads must land before LDS is reused
  __syncthreads();
#undef PH_ADVANCE
#undef PH_STAGE_A
#undef PH_STAGE_B
#undef PH_READ_A
#undef PH_READ_B
#undef PH_MMA

  // ---------------- epilogue: f32 math in registers -> 16-bit tile in LDS -> row-contiguous 16 B stores ----------
  const bool geglu = d.act == DBIR_ACT_GEGLU;
  const int N = d.N;
  const int bn_out = geglu ? PBN / 2 : PBN;
  const int cs_ld = bn_out + 8;
  u16* Cs = reinterpret_cast<u16*>(smem);
  const u16* __restrict__ RV = reinterpret_cast<const u16*>(d.rowvec);

  float4 b4[2][4];  // [nh][g] -> bias of 4 consecutive columns
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n0 = tn * PBN + wc * 64 + nh * 32 + 8 * g + 4 * hi;
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d.bias) {
        if (p.vec_bias && n0 + 4 <= N) {
          bb = *reinterpret_cast<const float4*>(d.bias + n0);
        } else {
          if (n0 + 0 < N) bb.x = d.bias[n0 + 0];
          if (n0 + 1 < N) bb.y = d.bias[n0 + 1];
          if (n0 + 2 < N) bb.z = d.bias[n0 + 2];
          if (n0 + 3 < N) bb.w = d.bias[n0 + 3];
        }
      }
      b4[nh][g] = bb;
    }

#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = mh * 128 + grp * 64 + i * 32 + lq;
      const int m = tm * PBM + row;
      const int mb = (m < M ? m : M - 1);
      const u16* rvp = RV ? RV + (long long)(mb / d.rows_per_batch) * d.rowvec_ld : nullptr;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        if (geglu && nh == 1) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wc * 64 + nh * 32 + 8 * g + 4 * hi;
          const int n0 = tn * PBN + nl;
          float v[4] = {acc[mh][nh][i][4 * g + 0] + b4[nh][g].x, acc[mh][nh][i][4 * g + 1] + b4[nh][g].y,
                        acc[mh][nh][i][4 * g + 2] + b4[nh][g].z, acc[mh][nh][i][4 * g + 3] + b4[nh][g].w};
          if (rvp) {
            if (p.vec_rv && n0 + 4 <= N) {
              const uint2 rr = *reinterpret_cast<const uint2*>(rvp + n0);
              v[0] += T::to_f32((u16)(rr.x & 0xffff));
              v[1] += T::to_f32((u16)(rr.x >> 16));
              v[2] += T::to_f32((u16)(rr.y & 0xffff));
              v[3] += T::to_f32((u16)(rr.y >> 16));
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n0 + e < N) v[e] += T::to_f32(rvp[n0 + e]);
            }
          }
          if (d.act == DBIR_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
          } else if (d.act == DBIR_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_fast_ph(v[e]);
          } else if (d.act == DBIR_ACT_LRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * d.act_param;
          } else if (geglu) {
            const float gt[4] = {acc[mh][1][i][4 * g + 0] + b4[1][g].x, acc[mh][1][i][4 * g + 1] + b4[1][g].y,
                                 acc[mh][1][i][4 * g + 2] + b4[1][g].z, acc[mh][1][i][4 * g + 3] + b4[1][g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= gelu_fast_ph(gt[e]);
          }
          const int ocl = geglu ? (wc * 32 + 8 * g + 4 * hi) : nl;
          uint2 pk;
          pk.x = (uint32_t)T::from_f32(v[0] * d.out_scale) | ((uint32_t)T::from_f32(v[1] * d.out_scale) << 16);
          pk.y = (uint32_t)T::from_f32(v[2] * d.out_scale) | ((uint32_t)T::from_f32(v[3] * d.out_scale) << 16);
          *reinterpret_cast<uint2*>(Cs + row * cs_ld + ocl) = pk;
        }
      }
    }
  __syncthreads();
  {
    const int n_out = geglu ? N / 2 : N;
    const int ch_per_row = bn_out >> 3;
    const int total = PBM * ch_per_row;
    const u16* __restrict__ Rg = d.R ? reinterpret_cast<const u16*>(d.R) + (long long)bz * d.strideR_z : nullptr;
    u16* __restrict__ Cg = reinterpret_cast<u16*>(d.C) + (long long)bz * d.strideC_z;
    for (int q = tid; q < total; q += PNT) {
      const int row = q / ch_per_row, ch = q - row * ch_per_row;
      const int m = tm * PBM + row;
      const int ncol = tn * bn_out + ch * 8;
      if (m >= M || ncol >= n_out) continue;
      uint4 v = *reinterpret_cast<const uint4*>(Cs + row * cs_ld + ch * 8);
      if (ncol + 8 <= n_out) {
        if (Rg) {
          const uint4 rr = *reinterpret_cast<const uint4*>(Rg + (long long)m * d.ldr + ncol);
          float a[8], bq[8];
          unpack8<T>(v, a);
          unpack8<T>(rr, bq);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += bq[e];
          v = pack8<T>(a);
        }
        *reinterpret_cast<uint4*>(Cg + (long long)m * d.ldc + ncol) = v;
      } else {
        float a[8];
        unpack8<T>(v, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (ncol + e < n_out) {
            float x = a[e];
            if (Rg) x += T::to_f32(Rg[(long long)m * d.ldr + ncol + e]);
            Cg[(long long)m * d.ldc + ncol + e] = T::from_f32(x);
          }
        }
      }
    }
  }
}

template <typename T>
int launch_ph(PhParams& p, hipStream_t s) {
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  auto kern = &gemm_ph_kernel<T>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS_BYTES);
    attr_set = true;
  }
  p.mtiles = cdiv(p.d.M, PBM);
  p.ntiles = cdiv(p.d.N, PBN);
  dim3 grid((unsigned)(p.mtiles * p.ntiles), p.d.batch > 0 ? p.d.batch : 1);
  hipLaunchKernelGGL(kern, grid, dim3(PNT), LDS_BYTES, s, p);
  DBIR_CHECK_LAUNCH("dbir_gemm(phased)");
  return DBIR_OK;
}

}  // namespace

// Same eligibility as the direct-to-LDS kernel (dbir_gemm_glds_eligible); called from dbir_gemm for tile 13.
int dbir_gemm_ph(const dbir_gemm_desc& dd, int Hv, int Wv, hipStream_t s) {
  PhParams p;
  p.d = dd;
  p.Hv = Hv;
  p.Wv = Wv;
  if (dd.mode == DBIR_MODE_LINEAR) {
    p.nkc = dd.K / BK;
    p.ntaps = 1;
  } else {
    p.nkc = dd.Cin / BK;
    p.ntaps = 9;
  }
  p.vec_bias = dd.bias && (reinterpret_cast<uintptr_t>(dd.bias) & 15) == 0;
  p.vec_rv = dd.rowvec && (reinterpret_cast<uintptr_t>(dd.rowvec) & 7) == 0 && dd.rowvec_ld % 4 == 0;
  static const int tap_inner = getenv("DBIR_TAP_INNER") ? atoi(getenv("DBIR_TAP_INNER")) : 1;  // as gemm_glds.hip
  p.tap_inner = tap_inner;
  static const int dbg = getenv("DBIR_GEMM_DEBUG") ? atoi(getenv("DBIR_GEMM_DEBUG")) : 0;
  p.debug = dbg;
  return dd.dtype == DBIR_F16 ? launch_ph<F16>(p, s) : launch_ph<BF16>(p, s);
}
