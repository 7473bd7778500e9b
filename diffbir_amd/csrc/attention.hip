// Flash-style attention forward for head_dim 64 on gfx950 MFMA (no mask), see include/dbir.h.
//
// Work decomposition: grid = (ceil(Lq/128), B*H); 256 threads = 4 wave64; each wave owns 32 query rows and
// walks the keys in tiles of 64 that the whole block stages through LDS.
//
// The score tile is computed TRANSPOSED, S^T = K Q^T (A-operand = K rows from LDS, B-operand = Q rows held
// in registers), so that with the 32x32x16 C/D layout (col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)) every
// lane holds 32 of the 64 scores of ONE query (the other 32 live in lane^32): the online-softmax row max / row
// sum are 31 in-register ops plus one cross-half shuffle, and the O^T accumulator rescale is lane-local.
//
// P feeds the second MFMA (O^T += V^T P^T) straight from the accumulator registers: for key-step s the
// B-operand element j of half `hi` is accumulator register 8*(s&1)+j of score block s>>1, which corresponds
// to key 16*s + 4*hi + (j&3) + 8*(j>>2).  The contraction index of an MFMA is arbitrary as long as both
// operands agree, so V^T is read from LDS with exactly that key permutation (two ds_read_b64 per step) and
// no cross-lane shuffles or LDS round trip for P are needed.  V arrives already transposed ([d][key], emitted
// by the producing GEMM's transposed store) so both LDS images are filled with coalesced 16-byte rows.
#include "common.h"

namespace {

// acc + lo(pk) + hi(pk) of a packed 16-bit pair: one v_dot2_f32_f16 / v_dot2_f32_bf16 against (1, 1)
template <typename T>
__device__ __forceinline__ float dot2_ones(uint32_t pk, float acc);
template <>
__device__ __forceinline__ float dot2_ones<F16>(uint32_t pk, float acc) {
  const f16x2 ones = {(f16)1.0f, (f16)1.0f};
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, pk), ones, acc, false);
}
template <>
__device__ __forceinline__ float dot2_ones<BF16>(uint32_t pk, float acc) {
  const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), ones, acc, false);
}

constexpr int KT = 64;        // keys per tile
constexpr int K_LD = 64 + 8;  // K tile row (halfs): 144 B -> conflict-free ds_read_b128
constexpr int V_LD = 64 + 4;  // V^T tile row (halfs): 136 B -> conflict-free ds_read_b64

// ---------------------------------------------------------------------------------------------------------------
// Self-attention kernel (any Lk): the decomposition and register dataflow described at the top of this file.
// At head_dim 64 a 64-key tile is only 16 MFMAs (512 matrix cycles) per wave against >300 VALU ops of a naive
// softmax, so the kernel is VALU-bound; hence:
//   * K / V^T tiles are double-buffered in LDS; the next tile's global loads are issued into registers BEFORE the
//     current tile's MFMAs and written to the other buffer after them (guide T14): one barrier per tile, L2/HBM
//     latency hidden under the compute.
//   * softmax VALU diet: the softmax scale is folded into the exponent (p = exp2(s*c - m), one v_fma + one v_exp
//     per score instead of mul, sub, exp), the row max is taken on raw scores (c > 0), the key mask only runs on a
//     ragged last tile, and the O / l rescale is skipped while the running max grows by less than 2^8 for every
//     row of the wave (guide T13; P is then bounded by 256, exact in f32 and well inside f16 / bf16 range; the
//     decision precedes the tile's exponentials and its P.V, the textbook order).
//   * FOLD (round 4, the production form): per-score work cut from 4.3 to 2.8 VALU instructions.  (i) Q is multiplied by
//     c = scale * log2(e) once per workgroup (f32 product, rounded once to the 16-bit operand) and the running maximum rides
//     into the score MFMA as its C operand — a 16-register block holding -m_run, rewritten only on a rescale — so the
//     accumulator comes out as s*c - m_run and p = exp2(acc): no v_fma per score.  (ii) the row sum is taken over the PACKED
//     16-bit P with v_dot2_f32_f16 / _bf16 against a vector of ones (one instruction per two scores; it is also the sum of
//     exactly the values the P.V MFMA multiplies).  The first tile sets m_run to its row maximum unconditionally; after
//     that the 2^8 rule is unchanged.
template <typename T, int OCC, bool FOLD = false>
__global__ __launch_bounds__(256, OCC) void attn2_kernel(const u16* __restrict__ Q, long long q_bs, long long ldq,
                                                       const u16* __restrict__ K, long long k_bs, long long ldk,
                                                       const u16* __restrict__ Vt, long long vt_bs, long long ldvt,
                                                       u16* __restrict__ O, long long o_bs, long long ldo, int H,
                                                       int Lq, int Lk, float c /* scale * log2(e) */) {
  constexpr int KS_HALFS = KT * K_LD, VS_HALFS = 64 * V_LD;
  __shared__ __attribute__((aligned(16))) u16 lds[2 * (KS_HALFS + VS_HALFS)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, lq = lane & 31;
  // 1-D grid of nqb * (B*H) blocks.  Workgroup w runs on XCD w % 8 (each XCD has its own L2): when B*H is a multiple
  // of 8 all query blocks of one (batch, head) are mapped to the same XCD, so its K / V^T (Lk x 64 x 2 x 2 bytes) are
  // fetched from HBM once instead of once per XCD (rocprofv3 FETCH_SIZE: 190 MB per launch before).
  const int nqb = (Lq + 127) / 128;
  int bh, qb;
  {
    const int w = blockIdx.x, BH = gridDim.x / nqb;
    if ((BH & 7) == 0) {
      const int xcd = w & 7, idx = w >> 3;
      bh = xcd + 8 * (idx / nqb);
      qb = idx % nqb;
    } else {
      bh = w / nqb;
      qb = w % nqb;
    }
  }
  const int b = bh / H, h = bh % H;
  const int q_row = qb * 128 + wave * 32 + lq;
  const bool q_ok = q_row < Lq;

  typename T::vec8 qf[4];
  {
    const u16* qp = Q + (long long)b * q_bs + (long long)(q_ok ? q_row : 0) * ldq + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 v = q_ok ? *reinterpret_cast<const uint4*>(qp + ks * 16) : make_uint4(0, 0, 0, 0);
      if (FOLD) {
        float f[8];
        unpack8<T>(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= c;
        v = pack8<T>(f);
      }
      qf[ks] = __builtin_bit_cast(typename T::vec8, v);
    }
  }
  f32x16 negm;  // FOLD: -m_run in every element = the C operand of the first MFMA of each score block
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  f32x16 o_acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[t][r] = 0.f;
  float m_run = FOLD ? 0.f : -1e30f, l_run = 0.f;  // m_run in the scaled log2 domain

  const u16* Kg = K + (long long)b * k_bs + h * 64;
  const u16* Vg = Vt + (long long)b * vt_bs + (long long)(h * 64) * ldvt;
  const int kc = tid & 7, r0 = tid >> 3;
  const int ntiles = (Lk + KT - 1) / KT;

  uint4 kreg[2], vreg[2];
  auto fetch = [&](int kt) {  // global -> registers (zero-filled outside [0, Lk))
    const int key0 = kt * KT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + 32 * i;
      const int key = key0 + row;
      kreg[i] = make_uint4(0, 0, 0, 0);
      if (key < Lk) kreg[i] = *reinterpret_cast<const uint4*>(Kg + (long long)key * ldk + kc * 8);
      const int kcol = key0 + kc * 8;
      uint4 vv = make_uint4(0, 0, 0, 0);
      if (kcol < Lk) vv = *reinterpret_cast<const uint4*>(Vg + (long long)row * ldvt + kcol);
      vreg[i] = vv;
    }
  };
  auto commit = [&](int buf, int kt) {  // registers -> LDS buffer (touching the loaded values only here keeps the
                                        // loads asynchronous: no vmcnt wait before this point)
    u16* Ks = lds + buf * (KS_HALFS + VS_HALFS);
    u16* Vs = Ks + KS_HALFS;
    const int kcol = kt * KT + kc * 8;
    if (kt * KT + KT > Lk) {  // ragged last tile: pad columns of V^T may hold anything -> zero keys >= Lk
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u16* hv = reinterpret_cast<u16*>(&vreg[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (kcol + e >= Lk) hv[e] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + 32 * i;
      *reinterpret_cast<uint4*>(&Ks[row * K_LD + kc * 8]) = kreg[i];
      uint2* dst = reinterpret_cast<uint2*>(&Vs[row * V_LD + kc * 8]);
      dst[0] = make_uint2(vreg[i].x, vreg[i].y);
      dst[1] = make_uint2(vreg[i].z, vreg[i].w);
    }
  };

  fetch(0);
  commit(0, 0);
  __syncthreads();

#ifdef DBIR_DIAG  // tile-loop anatomy (tools/probes/attn_diag.py): s_memtime accumulators per section
  unsigned long long ta[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define ATS(I)                                                    \
  do {                                                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    ta[I] += now_ - tprev;                                        \
    tprev = now_;                                                 \
  } while (0)
#else
#define ATS(I)
#endif
  for (int kt = 0; kt < ntiles; ++kt) {
    const int key0 = kt * KT;
    const u16* Ks = lds + (kt & 1) * (KS_HALFS + VS_HALFS);
    const u16* Vs = Ks + KS_HALFS;
    const bool more = kt + 1 < ntiles;
    if (more) fetch(kt + 1);  // in flight during this tile's MFMAs
    ATS(0);

    f32x16 s_acc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (!FOLD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        typename T::vec8 kf =
            *reinterpret_cast<const typename T::vec8*>(&Ks[(kb * 32 + lq) * K_LD + ks * 16 + hi * 8]);
        if (FOLD && ks == 0) s_acc[kb] = T::mfma32(kf, qf[0], negm);
        else s_acc[kb] = T::mfma32(kf, qf[ks], s_acc[kb]);
      }
    }
    ATS(1);
    if (key0 + KT > Lk) {  // ragged last tile: mask keys >= Lk (wave-uniform branch)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= Lk) s_acc[kb][r] = -1e30f;
        }
    }
    float mx = s_acc[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float psum = 0.f;
    if (FOLD) {
      // the accumulators hold s*c - m_run.  First tile: m_run := the row maximum; later: the 2^8 rule of the header
      if (kt == 0 || !__all(mx <= 8.0f)) {
        const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
        m_run += delta;
        if (kt != 0) {  // (o_acc = l_run = 0 in the first tile, and exp2(-delta) may overflow there)
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run *= alpha;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[t][r] *= alpha;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s_acc[kb][r] -= delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_run;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[kb][r] = __builtin_amdgcn_exp2f(s_acc[kb][r]);
    } else {
      const float mxs = mx * c;
      if (!__all(mxs - m_run <= 8.0f)) {  // some row's max grew by more than 2^8: rescale everything once
        const float m_new = fmaxf(m_run, mxs);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o_acc[t][r] *= alpha;
      }
      const float neg_m = -m_run;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][r], c, neg_m));
          s_acc[kb][r] = pv;
          psum += pv;
        }
    }
    ATS(2);

#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float pf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = s_acc[s >> 1][8 * (s & 1) + j];
      const uint4 pp = pack8<T>(pf);
      if (FOLD) psum = dot2_ones<T>(pp.w, dot2_ones<T>(pp.z, dot2_ones<T>(pp.y, dot2_ones<T>(pp.x, psum))));
      const typename T::vec8 pfrag = __builtin_bit_cast(typename T::vec8, pp);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const u16* vrow = &Vs[(t * 32 + lq) * V_LD + 16 * s + 4 * hi];
        const uint2 v0 = *reinterpret_cast<const uint2*>(vrow);
        const uint2 v1 = *reinterpret_cast<const uint2*>(vrow + 8);
        const uint4 vv = make_uint4(v0.x, v0.y, v1.x, v1.y);
        o_acc[t] = T::mfma32(__builtin_bit_cast(typename T::vec8, vv), pfrag, o_acc[t]);
      }
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run += psum;
    ATS(3);
    if (more) commit((kt + 1) & 1, kt + 1);  // the other buffer was last read in iteration kt-1 (barrier since)
    ATS(4);
    __syncthreads();
    ATS(5);
  }
#ifdef DBIR_DIAG
  // DIAG build only: the accumulators replace the first output row of this wave's 32-row slab — the harness
  // (tools/probes/attn_diag.py) reads them back from O and does not look at the attention result.  s_memtime does not order
  // vector instructions, so only the fetch / commit / barrier sections are reliable; MFMA + softmax issue is the rest.
  if (lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(O + (long long)b * o_bs +
                                                                 (long long)(qb * 128 + wave * 32) * ldo + h * 64);
    for (int i = 0; i < 6; ++i) o[i] = ta[i];
    o[6] = ntiles;
  }
  return;
#endif
#undef ATS

  if (q_ok) {
    const float inv = 1.0f / l_run;
    u16* op = O + (long long)b * o_bs + (long long)q_row * ldo + h * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u16 hv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = T::from_f32(o_acc[t][4 * g + e] * inv);
        uint2 pk;
        pk.x = (uint32_t)hv[0] | ((uint32_t)hv[1] << 16);
        pk.y = (uint32_t)hv[2] | ((uint32_t)hv[3] << 16);
        *reinterpret_cast<uint2*>(op + t * 32 + 8 * g + 4 * hi) = pk;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// attn3_kernel (round 6): the same arithmetic as attn2_kernel<T, *, FOLD = true> with EIGHT waves per workgroup (256 query
// rows) whose two wave groups (waves 0-3 / 4-7 = one wave of each on every SIMD) run ONE BARRIER APART.
// Why: per 64-key tile a wave issues 16 MFMAs (512 matrix cycles) and ~740 cycles of softmax VALU work, and on this chip
// the VALU and MFMA work of ONE wave do not overlap (profiles/r3_mfma_valu_overlap.txt, r6_ff64_gateA_v1.txt: a wave's own
// MFMAs hide none of its VALU instructions) while two waves of a SIMD overlap fully IF one is in its matrix phase while the
// other is in its VALU phase.  Independent workgroups (attn2: 2 - 3 resident per CU) drift in and out of that alignment:
// 1400 SIMD cycles per tile measured = the serial sum.  Here the alignment is constructed: every wave alternates
//   matrix part : O^T += V^T P^T of tile t - 1, S^T = K Q^T of tile t, write its share of tile t + 1 into LDS, fetch t + 2
//   barrier
//   VALU part   : softmax of tile t (running maximum, exponentials, packing, row sum)
//   barrier
// and group 1 executes one barrier more in front of the loop (group 0 one behind it): in every barrier interval one wave
// of each SIMD multiplies while its partner exponentiates.  K / V^T tiles live in a 3-deep LDS ring (a tile is read by the
// two groups in consecutive intervals; its slot is rewritten two tiles later).
template <typename T>
__global__ __launch_bounds__(512) void attn3_kernel(const u16* __restrict__ Q, long long q_bs, long long ldq,
                                                    const u16* __restrict__ K, long long k_bs, long long ldk,
                                                    const u16* __restrict__ Vt, long long vt_bs, long long ldvt,
                                                    u16* __restrict__ O, long long o_bs, long long ldo, int H, int Lq,
                                                    int Lk, float c /* scale * log2(e) */) {
  constexpr int KS_HALFS = KT * K_LD, VS_HALFS = 64 * V_LD, NB = 3;
  __shared__ __attribute__((aligned(16))) u16 lds[NB * (KS_HALFS + VS_HALFS)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int hi = lane >> 5, lq = lane & 31;
  const int nqb = (Lq + 255) / 256;
  int bh, qb;
  {
    const int w = blockIdx.x, BH = gridDim.x / nqb;
    if ((BH & 7) == 0) {  // all query blocks of one (batch, head) on one XCD: its K / V^T are fetched from HBM once
      const int xcd = w & 7, idx = w >> 3;
      bh = xcd + 8 * (idx / nqb);
      qb = idx % nqb;
    } else {
      bh = w / nqb;
      qb = w % nqb;
    }
  }
  const int b = bh / H, h = bh % H;
  const int q_row = qb * 256 + wave * 32 + lq;
  const bool q_ok = q_row < Lq;

  typename T::vec8 qf[4];
  {
    const u16* qp = Q + (long long)b * q_bs + (long long)(q_ok ? q_row : 0) * ldq + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 v = q_ok ? *reinterpret_cast<const uint4*>(qp + ks * 16) : make_uint4(0, 0, 0, 0);
      float f[8];
      unpack8<T>(v, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= c;
      qf[ks] = __builtin_bit_cast(typename T::vec8, pack8<T>(f));
    }
  }
  f32x16 negm;  // -m_run in every element = the C operand of the first MFMA of each score block
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  f32x16 o_acc[2], s_acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[t][r] = 0.f;
  float m_run = 0.f, l_run = 0.f;
  uint4 pp[4];  // the tile's probabilities, packed: B operands of the four key-steps of O^T += V^T P^T

  const u16* Kg = K + (long long)b * k_bs + h * 64;
  const u16* Vg = Vt + (long long)b * vt_bs + (long long)(h * 64) * ldvt;
  const int kc = tid & 7, row = tid >> 3;  // one 16-byte chunk of the K tile and one of the V^T tile per thread
  const int ntiles = (Lk + KT - 1) / KT;
  uint4 kreg, vreg;
  auto fetch = [&](int kt) {
    const int key0 = kt * KT;
    kreg = make_uint4(0, 0, 0, 0);
    if (key0 + row < Lk) kreg = *reinterpret_cast<const uint4*>(Kg + (long long)(key0 + row) * ldk + kc * 8);
    vreg = make_uint4(0, 0, 0, 0);
    if (key0 + kc * 8 < Lk) vreg = *reinterpret_cast<const uint4*>(Vg + (long long)row * ldvt + key0 + kc * 8);
  };
  auto commit = [&](int kt) {
    u16* Ks = lds + (kt % NB) * (KS_HALFS + VS_HALFS);
    u16* Vs = Ks + KS_HALFS;
    const int kcol = kt * KT + kc * 8;
    if (kt * KT + KT > Lk) {  // ragged last tile: pad columns of V^T may hold anything -> zero keys >= Lk
      u16* hv = reinterpret_cast<u16*>(&vreg);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (kcol + e >= Lk) hv[e] = 0;
    }
    *reinterpret_cast<uint4*>(&Ks[row * K_LD + kc * 8]) = kreg;
    uint2* dst = reinterpret_cast<uint2*>(&Vs[row * V_LD + kc * 8]);
    dst[0] = make_uint2(vreg.x, vreg.y);
    dst[1] = make_uint2(vreg.z, vreg.w);
  };
  auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto pv = [&](int kt) {  // O^T += V^T P^T of tile kt
    const u16* Vs = lds + (kt % NB) * (KS_HALFS + VS_HALFS) + KS_HALFS;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4 p4 = pp[s];
      const typename T::vec8 pfrag = __builtin_bit_cast(typename T::vec8, p4);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const u16* vrow = &Vs[(t * 32 + lq) * V_LD + 16 * s + 4 * hi];
        const uint2 v0 = *reinterpret_cast<const uint2*>(vrow);
        const uint2 v1 = *reinterpret_cast<const uint2*>(vrow + 8);
        const uint4 vv = make_uint4(v0.x, v0.y, v1.x, v1.y);
        o_acc[t] = T::mfma32(__builtin_bit_cast(typename T::vec8, vv), pfrag, o_acc[t]);
      }
    }
  };

  fetch(0);
  commit(0);
  if (ntiles > 1) fetch(1);
  bar();
  if (grp) bar();  // group 1 runs one barrier behind group 0
  for (int kt = 0; kt < ntiles; ++kt) {
    // ---------------- matrix part ----------------
    if (kt > 0) pv(kt - 1);
    {
      const u16* Ks = lds + (kt % NB) * (KS_HALFS + VS_HALFS);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          typename T::vec8 kf = *reinterpret_cast<const typename T::vec8*>(&Ks[(kb * 32 + lq) * K_LD + ks * 16 + hi * 8]);
          if (ks == 0) s_acc[kb] = T::mfma32(kf, qf[0], negm);
          else s_acc[kb] = T::mfma32(kf, qf[ks], s_acc[kb]);
        }
    }
    if (kt + 1 < ntiles) {
      commit(kt + 1);
      if (kt + 2 < ntiles) fetch(kt + 2);
    }
    bar();
    // ---------------- VALU part ----------------
    const int key0 = kt * KT;
    if (key0 + KT > Lk) {  // ragged last tile: mask keys >= Lk (wave-uniform branch)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= Lk) s_acc[kb][r] = -1e30f;
        }
    }
    float mx = s_acc[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // the accumulators hold s*c - m_run.  First tile: m_run := the row maximum; later: rescale only when some row's maximum
    // grew by more than 2^8 (attn2_kernel's rule)
    if (kt == 0 || !__all(mx <= 8.0f)) {
      const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
      m_run += delta;
      if (kt != 0) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l_run *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o_acc[t][r] *= alpha;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[kb][r] -= delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) negm[r] = -m_run;
    }
    float psum = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float pf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = __builtin_amdgcn_exp2f(s_acc[s >> 1][8 * (s & 1) + j]);
      pp[s] = pack8<T>(pf);
      psum = dot2_ones<T>(pp[s].w, dot2_ones<T>(pp[s].z, dot2_ones<T>(pp[s].y, dot2_ones<T>(pp[s].x, psum))));
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run += psum;
    bar();
  }
  pv(ntiles - 1);
  if (!grp) bar();

  if (q_ok) {
    const float inv = 1.0f / l_run;
    u16* op = O + (long long)b * o_bs + (long long)q_row * ldo + h * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u16 hv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = T::from_f32(o_acc[t][4 * g + e] * inv);
        uint2 pk;
        pk.x = (uint32_t)hv[0] | ((uint32_t)hv[1] << 16);
        pk.y = (uint32_t)hv[2] | ((uint32_t)hv[3] << 16);
        *reinterpret_cast<uint2*>(op + t * 32 + 8 * g + 4 * hi) = pk;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Cross-attention kernel (Lk <= 96: the 77-token text context, reference attention.py:189-216 with context != x).
// The whole K [Lk][64] and V^T [64][Lk] of one (batch, head) fit in 26 KB of LDS: they are staged ONCE per workgroup,
// which then walks `qpb` consecutive 128-query blocks with no further barrier; all keys sit in one tile, so the softmax
// is a single exact pass (no running max / rescale) — S^T = K Q^T (3 key blocks x 4 d-steps), p = exp2(s*c - m),
// O^T = V^T P^T (6 key-steps x 2 d-tiles), same register dataflow as attn2_kernel.  The generic kernel re-stages K / V^T
// through registers for every 128-query block behind two barriers: 154-185 TF/s at Lk = 77 (profiles/r1_attention_ab*).
constexpr int XK = 96;             // padded key count
constexpr int XV_LD = XK + 4;      // V^T row (halfs): 200 B -> conflict-free ds_read_b64

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_cross_kernel(const u16* __restrict__ Q, long long q_bs, long long ldq,
                                                            const u16* __restrict__ K, long long k_bs, long long ldk,
                                                            const u16* __restrict__ Vt, long long vt_bs, long long ldvt,
                                                            u16* __restrict__ O, long long o_bs, long long ldo, int H,
                                                            int Lq, int Lk, float c, int qpb) {
  __shared__ __attribute__((aligned(16))) u16 lds[XK * K_LD + 64 * XV_LD];
  u16* Ks = lds;
  u16* Vs = lds + XK * K_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, lq = lane & 31;
  const int nqb = (Lq + 127) / 128, ngrp = (nqb + qpb - 1) / qpb;
  const int bh = blockIdx.x / ngrp, grp = blockIdx.x % ngrp;
  const int b = bh / H, h = bh % H;
  const u16* Kg = K + (long long)b * k_bs + h * 64;
  const u16* Vg = Vt + (long long)b * vt_bs + (long long)(h * 64) * ldvt;
  // ---- stage K (rows >= Lk zero) and V^T (columns >= Lk zero) once ----
  for (int q = tid; q < XK * 8; q += 256) {
    const int row = q >> 3, kc = q & 7;
    uint4 kv = make_uint4(0, 0, 0, 0);
    if (row < Lk) kv = *reinterpret_cast<const uint4*>(Kg + (long long)row * ldk + kc * 8);
    *reinterpret_cast<uint4*>(&Ks[row * K_LD + kc * 8]) = kv;
  }
  for (int q = tid; q < 64 * (XK / 8); q += 256) {
    const int row = q / (XK / 8), kc = q % (XK / 8);
    const int kcol = kc * 8;
    uint4 vv = make_uint4(0, 0, 0, 0);
    if (kcol < Lk) {
      vv = *reinterpret_cast<const uint4*>(Vg + (long long)row * ldvt + kcol);
      if (kcol + 8 > Lk) {
        u16* hv = reinterpret_cast<u16*>(&vv);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (kcol + e >= Lk) hv[e] = 0;
      }
    }
    uint2* dst = reinterpret_cast<uint2*>(&Vs[row * XV_LD + kcol]);
    dst[0] = make_uint2(vv.x, vv.y);
    dst[1] = make_uint2(vv.z, vv.w);
  }
  __syncthreads();

  for (int qi = 0; qi < qpb; ++qi) {
    const int qb = grp * qpb + qi;
    if (qb >= nqb) break;
    const int q_row = qb * 128 + wave * 32 + lq;
    const bool q_ok = q_row < Lq;
    typename T::vec8 qf[4];
    {
      const u16* qp = Q + (long long)b * q_bs + (long long)(q_ok ? q_row : 0) * ldq + h * 64 + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        uint4 v = q_ok ? *reinterpret_cast<const uint4*>(qp + ks * 16) : make_uint4(0, 0, 0, 0);
        qf[ks] = __builtin_bit_cast(typename T::vec8, v);
      }
    }
    f32x16 s_acc[3];
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        typename T::vec8 kf =
            *reinterpret_cast<const typename T::vec8*>(&Ks[(kb * 32 + lq) * K_LD + ks * 16 + hi * 8]);
        s_acc[kb] = T::mfma32(kf, qf[ks], s_acc[kb]);
      }
    }
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 3; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float sv = key < Lk ? s_acc[kb][r] : -1e30f;
        s_acc[kb][r] = sv;
        mx = fmaxf(mx, sv);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float neg_m = -mx * c;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 3; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][r], c, neg_m));  // masked keys: exp2(-huge) = 0
        s_acc[kb][r] = pv;
        psum += pv;
      }
    psum += __shfl_xor(psum, 32, 64);
    f32x16 o_acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      float pf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = s_acc[s >> 1][8 * (s & 1) + j];
      const uint4 pp = pack8<T>(pf);
      const typename T::vec8 pfrag = __builtin_bit_cast(typename T::vec8, pp);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const u16* vrow = &Vs[(t * 32 + lq) * XV_LD + 16 * s + 4 * hi];
        const uint2 v0 = *reinterpret_cast<const uint2*>(vrow);
        const uint2 v1 = *reinterpret_cast<const uint2*>(vrow + 8);
        const uint4 vv = make_uint4(v0.x, v0.y, v1.x, v1.y);
        o_acc[t] = T::mfma32(__builtin_bit_cast(typename T::vec8, vv), pfrag, o_acc[t]);
      }
    }
    if (q_ok) {
      const float inv = 1.0f / psum;
      u16* op = O + (long long)b * o_bs + (long long)q_row * ldo + h * 64;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u16 hv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = T::from_f32(o_acc[t][4 * g + e] * inv);
          uint2 pk;
          pk.x = (uint32_t)hv[0] | ((uint32_t)hv[1] << 16);
          pk.y = (uint32_t)hv[2] | ((uint32_t)hv[3] << 16);
          *reinterpret_cast<uint2*>(op + t * 32 + 8 * g + 4 * hi) = pk;
        }
    }
  }
}

int g_attn_variant = 2;
int g_attn3_min_lq = 512;   // (env DBIR_ATTN3_MIN_LQ at load time: A/B of the dispatch threshold)

}  // namespace

// A/B switch (dbir_set_option): 2 = default (cross kernel for Lk <= 96, generic otherwise), 3 = generic kernel always,
// 8 = the 8-wave attn3_kernel for long self-attentions (experiment, slower); 4 / 5 / 6 = default dispatch with the generic kernel's pre-round-4 softmax (FOLD = false) compiled for 4 / 3 / default resident
// waves per SIMD; 2 and 3 run the FOLD form (scale folded into Q, running maximum as the score MFMA's C operand, row sum by v_dot2)
void dbir_attention_set_variant(int v) {
  if (v >= 1000) g_attn3_min_lq = v - 1000;   // 1000 + n: threshold of the 8-wave kernel (rows of Q), variant unchanged
  else g_attn_variant = v;
}

extern "C" int dbir_attention(int dtype, const void* Q, long long q_bstride, long long ldq, const void* K,
                              long long k_bstride, long long ldk, const void* Vt, long long vt_bstride,
                              long long ldvt, void* O, long long o_bstride, long long ldo, int B, int H, int Lq,
                              int Lk, float scale, void* stream) {
  DBIR_CHECK_ARG(Q && K && Vt && O, "dbir_attention: null pointer");
  DBIR_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lk > 0, "dbir_attention: bad sizes");
  DBIR_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0 && q_bstride % 8 == 0 &&
                     k_bstride % 8 == 0 && vt_bstride % 8 == 0 && o_bstride % 4 == 0,
                 "dbir_attention: strides must keep 16-byte (Q,K,Vt) / 8-byte (O) alignment");
  DBIR_CHECK_ARG(ldvt >= ((Lk + 7) / 8) * 8, "dbir_attention: ldvt %lld too small for Lk %d", ldvt, Lk);
  DBIR_CHECK_ARG((long long)B * H <= 65535, "dbir_attention: B*H too large for gridDim.y");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const float sl2 = scale * 1.4426950408889634f;
  if (dtype != DBIR_F16 && dtype != DBIR_BF16) {
    dbir_set_error("dbir_attention: bad dtype %d", dtype);
    return DBIR_ERR_ARG;
  }
  // text-context cross-attention with K / V^T resident in LDS (3 = A/B: generic kernel).  At Lk = 77 the op is HBM-bound
  // (Q in + O out = 84 MB for 6.5 GFLOP at the 64x64 level: 77 FLOP/B against a ridge of ~400): measured 29.3 vs 34.7 us
  // there (2.9 TB/s), but below ~2048 query blocks the generic kernel's finer grid wins (profiles/r2_attention_ab.log)
  if (Lk <= XK && g_attn_variant != 3 && (long long)cdiv(Lq, 128) * B * H >= 2048) {
    const int nqb = cdiv(Lq, 128);
    // enough workgroups to fill 256 CUs x 2, each amortising its K / V^T staging over up to 8 query blocks
    int qpb = 8;
    while (qpb > 1 && (long long)cdiv(nqb, qpb) * B * H < 1024) qpb >>= 1;
    const dim3 gridx((unsigned)(cdiv(nqb, qpb) * B * H));
    if (dtype == DBIR_F16)
      hipLaunchKernelGGL((attn_cross_kernel<F16>), gridx, dim3(256), 0, s, (const u16*)Q, q_bstride, ldq, (const u16*)K,
                         k_bstride, ldk, (const u16*)Vt, vt_bstride, ldvt, (u16*)O, o_bstride, ldo, H, Lq, Lk, sl2, qpb);
    else
      hipLaunchKernelGGL((attn_cross_kernel<BF16>), gridx, dim3(256), 0, s, (const u16*)Q, q_bstride, ldq, (const u16*)K,
                         k_bstride, ldk, (const u16*)Vt, vt_bstride, ldvt, (u16*)O, o_bstride, ldo, H, Lq, Lk, sl2, qpb);
    DBIR_CHECK_LAUNCH("dbir_attention(cross)");
    return DBIR_OK;
  }
  DBIR_CHECK_ARG((long long)cdiv(Lq, 128) * B * H < 2147483647LL, "dbir_attention: grid too large");
  // round 6 experiment, OFF by default (variant 8 = on): the 8-wave kernel whose wave groups alternate matrix / softmax phases —
  // measured 0.65x of attn2 (profiles/r6_attn3_ab.txt)
  if (g_attn_variant == 8 && Lq >= g_attn3_min_lq && Lk >= 256 && (long long)cdiv(Lq, 256) * B * H >= 256) {
    const dim3 grid3((unsigned)(cdiv(Lq, 256) * B * H));
    if (dtype == DBIR_F16)
      hipLaunchKernelGGL((attn3_kernel<F16>), grid3, dim3(512), 0, s, (const u16*)Q, q_bstride, ldq, (const u16*)K, k_bstride, ldk,
                         (const u16*)Vt, vt_bstride, ldvt, (u16*)O, o_bstride, ldo, H, Lq, Lk, sl2);
    else
      hipLaunchKernelGGL((attn3_kernel<BF16>), grid3, dim3(512), 0, s, (const u16*)Q, q_bstride, ldq, (const u16*)K, k_bstride, ldk,
                         (const u16*)Vt, vt_bstride, ldvt, (u16*)O, o_bstride, ldo, H, Lq, Lk, sl2);
    DBIR_CHECK_LAUNCH("dbir_attention(attn3)");
    return DBIR_OK;
  }
  const dim3 grid1((unsigned)(cdiv(Lq, 128) * B * H));
#define ATTN2_LAUNCH(TT, OCC, ...)                                                                                     \
  hipLaunchKernelGGL((attn2_kernel<TT, OCC, ##__VA_ARGS__>), grid1, dim3(256), 0, s, (const u16*)Q, q_bstride, ldq,    \
                     (const u16*)K, k_bstride, ldk, (const u16*)Vt, vt_bstride, ldvt, (u16*)O, o_bstride, ldo, H, Lq,  \
                     Lk, sl2)
  // register budget variants (A/B through dbir_set_option): 2 waves / SIMD guaranteed (the compiler lands on 148 VGPRs for
  // the FOLD form, 133 for the older one = 3 resident), 4: capped at 128 VGPRs = 4 resident, 5: 3 resident by construction
  const int occ = g_attn_variant == 4 ? 4 : (g_attn_variant == 5 ? 3 : 2);
  if (dtype == DBIR_F16) {
    if (occ == 4) ATTN2_LAUNCH(F16, 4);
    else if (occ == 3) ATTN2_LAUNCH(F16, 3);
    else if (g_attn_variant == 6) ATTN2_LAUNCH(F16, 2);
    else ATTN2_LAUNCH(F16, 2, true);
  } else {
    if (occ == 4) ATTN2_LAUNCH(BF16, 4);
    else if (occ == 3) ATTN2_LAUNCH(BF16, 3);
    else if (g_attn_variant == 6) ATTN2_LAUNCH(BF16, 2);
    else ATTN2_LAUNCH(BF16, 2, true);
  }
#undef ATTN2_LAUNCH
  DBIR_CHECK_LAUNCH("dbir_attention");
  return DBIR_OK;
}
