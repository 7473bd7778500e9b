// Fused implicit-GEMM (linear / 1x1 / 3x3 convolution) on gfx950 MFMA.
//
//   C[M,N] = epilogue( sum_k A(m,k) * W[n,k] )        (see include/dbir.h: dbir_gemm)
//
// Structure (round 1: correctness-first, LDS-tiled, register-prefetch double buffering):
//   * 256 threads = 4 wave64 in a 2x2 grid; each wave owns (32*MI) x (32*NJ) outputs as MI x NJ
//     v_mfma_f32_32x32x16 accumulator tiles (f32x16 each)  -> block tile (64*MI) x (64*NJ), BK = 64.
//   * A and W tiles are staged global -> VGPR (16 B per lane, coalesced along K) -> LDS with rows padded to
//     72 halfs (144 B): the per-lane-group ds_read_b128 fragment reads are then bank-conflict free
//     (bank = (row*36 + 4*hi) mod 64 distinct for the 16 lanes of each b128 service group).
//   * The next K-tile's global loads are issued before the current tile's MFMAs and written to the other LDS
//     buffer afterwards: one __syncthreads per K-tile, HBM/L2 latency hidden under 16*MI*NJ/... MFMAs.
//   * conv3x3 is an implicit GEMM: the A-tile "row" is an output pixel, the K index walks (tap, channel);
//     padding, stride 2, the VAE's asymmetric pad and a fused nearest-x2 upsample are address math on the gather.
//   * epilogue fused in registers: bias, per-sample row vector (time-embedding add), SiLU/GELU/LeakyReLU/GEGLU,
//     scale, residual add, f16/bf16/f32 store or per-batch transposed store (V^T for attention).
//
// MFMA fragment layout used (cdna_hip_programming.md §3): for v_mfma_f32_32x32x16_{f16,bf16}
//   A operand: lane l holds A[i = l&31][k = 8*(l>>5) .. +7];  B operand: lane l holds B[k = 8*(l>>5)..+7][j = l&31]
//   C/D: lane l, reg r -> row i = (r&3) + 8*(r>>2) + 4*(l>>5), col j = l&31.
// We feed A-operand = activation rows (m), B-operand = weight rows (n), so D rows = m, D cols = n.
#include "common.h"

namespace {

constexpr int BK = 64;
constexpr int LDS_LD = BK + 8;  // halfs per LDS row (144 B)

struct GemmParams {
  dbir_gemm_desc d;
  int Hv, Wv;  // virtual (upsampled) input extent for conv bounds checks
};

template <typename T, int MI, int NJ>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int BM = 64 * MI, BN = 64 * NJ;
  constexpr int A_IT = BM * 8 / 256;  // 16-byte chunks of the A tile per thread
  constexpr int B_IT = BN * 8 / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u16* As = reinterpret_cast<u16*>(smem);      // [2][BM][LDS_LD]
  u16* Bs = As + 2 * BM * LDS_LD;              // [2][BN][LDS_LD]

  const dbir_gemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bm = blockIdx.x, bn = blockIdx.y, bz = blockIdx.z;
  const int M = d.M, N = d.N, K = d.K;

  const u16* __restrict__ Ag = reinterpret_cast<const u16*>(d.A) + (long long)bz * d.strideA_z;
  const u16* __restrict__ Wg = reinterpret_cast<const u16*>(d.W) + (long long)bz * d.strideW_z;

  const int kc = tid & 7;   // chunk index along K within the BK tile
  const int r0 = tid >> 3;  // 0..31: tile row handled (plus 32*i)

  // ---- per-row precompute for the A gather ----
  long long a_base[A_IT];  // linear: row offset; conv: batch offset
  int a_iy0[A_IT], a_ix0[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = bm * BM + r0 + 32 * i;
    a_ok[i] = m < M;
    if (d.mode == DBIR_MODE_LINEAR) {
      a_base[i] = (long long)m * d.lda;
      a_iy0[i] = a_ix0[i] = 0;
    } else {
      const int hw = d.Ho * d.Wo;
      const int b = m / hw;
      const int rem = m - b * hw;
      const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
      a_base[i] = (long long)b * d.Hi * d.Wi * d.Cin;
      a_iy0[i] = oy * d.stride - d.pad;
      a_ix0[i] = ox * d.stride - d.pad;
    }
  }
  // conv K-walk state for this thread's chunk: k = kt*BK + kc*8 -> (tap, c)
  int tap = 0, cch = kc * 8;
  if (d.mode == DBIR_MODE_CONV3X3) {
    tap = cch / d.Cin;
    cch -= tap * d.Cin;
  }
  // W rows handled by this thread
  long long w_off[B_IT];
  bool w_ok[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = bn * BN + r0 + 32 * i;
    w_ok[i] = n < d.Wrows;
    w_off[i] = (long long)n * d.Kpad + kc * 8;
  }

  uint4 ra[A_IT], rb[B_IT];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

// Global -> register staging of K-tile `kt` (macro, not a lambda: by-reference lambda captures of the
// staging arrays defeat SROA and push them to scratch memory).
#define LOAD_TILE(kt_)                                                                                        \
  do {                                                                                                        \
    if (d.mode == DBIR_MODE_LINEAR) {                                                                         \
      const int k_ = (kt_) * BK + kc * 8;                                                                     \
      const bool kok_ = k_ < K;                                                                               \
      _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                                      \
        uint4 v_ = zero4;                                                                                     \
        if (a_ok[i] && kok_) v_ = *reinterpret_cast<const uint4*>(Ag + a_base[i] + k_);                       \
        ra[i] = v_;                                                                                           \
      }                                                                                                       \
    } else {                                                                                                  \
      const bool kok_ = tap < 9;                                                                              \
      const int ky_ = (tap * 11) >> 5, kx_ = tap - 3 * ky_;                                                   \
      _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                                      \
        int iy_ = a_iy0[i] + ky_, ix_ = a_ix0[i] + kx_;                                                       \
        const bool ok_ = a_ok[i] && kok_ && iy_ >= 0 && iy_ < p.Hv && ix_ >= 0 && ix_ < p.Wv;                 \
        if (d.upsample) {                                                                                     \
          iy_ >>= 1;                                                                                          \
          ix_ >>= 1;                                                                                          \
        }                                                                                                     \
        uint4 v_ = zero4;                                                                                     \
        if (ok_) v_ = *reinterpret_cast<const uint4*>(Ag + a_base[i] + ((long long)iy_ * d.Wi + ix_) * d.Cin + cch); \
        ra[i] = v_;                                                                                           \
      }                                                                                                       \
      cch += BK;                                                                                              \
      if (d.Cin >= BK) {                                                                                      \
        if (cch >= d.Cin) {                                                                                   \
          cch -= d.Cin;                                                                                       \
          ++tap;                                                                                              \
        }                                                                                                     \
      } else {                                                                                                \
        const int q_ = cch / d.Cin;                                                                           \
        tap += q_;                                                                                            \
        cch -= q_ * d.Cin;                                                                                    \
      }                                                                                                       \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                                        \
      uint4 v_ = zero4;                                                                                       \
      if (w_ok[i]) v_ = *reinterpret_cast<const uint4*>(Wg + w_off[i] + (long long)(kt_) * BK);               \
      rb[i] = v_;                                                                                             \
    }                                                                                                         \
  } while (0)
#define STORE_TILE(buf_)                                                                                     \
  do {                                                                                                       \
    u16* a_ = As + (buf_) * BM * LDS_LD;                                                                     \
    u16* b_ = Bs + (buf_) * BN * LDS_LD;                                                                     \
    _Pragma("unroll") for (int i = 0; i < A_IT; ++i)                                                         \
        *reinterpret_cast<uint4*>(a_ + (r0 + 32 * i) * LDS_LD + kc * 8) = ra[i];                             \
    _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                                         \
        *reinterpret_cast<uint4*>(b_ + (r0 + 32 * i) * LDS_LD + kc * 8) = rb[i];                             \
  } while (0)

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (K + BK - 1) / BK;
  LOAD_TILE(0);
  STORE_TILE(0);
  __syncthreads();

  const int frow = lane & 31, fk = (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) LOAD_TILE(kt + 1);
    const u16* a = As + cur * BM * LDS_LD + (wm * 32 * MI + frow) * LDS_LD + fk;
    const u16* b = Bs + cur * BN * LDS_LD + (wn * 32 * NJ + frow) * LDS_LD + fk;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      typename T::vec8 af[MI], bf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const typename T::vec8*>(a + i * 32 * LDS_LD + ks * 16);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        bf[j] = *reinterpret_cast<const typename T::vec8*>(b + j * 32 * LDS_LD + ks * 16);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma32(af[i], bf[j], acc[i][j]);
    }
    if (kt + 1 < nk) STORE_TILE(cur ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const u16* __restrict__ Rg = d.R ? reinterpret_cast<const u16*>(d.R) + (long long)bz * d.strideR_z : nullptr;
  const u16* __restrict__ RV = reinterpret_cast<const u16*>(d.rowvec);
  const int hi = lane >> 5;
  const bool geglu = d.act == DBIR_ACT_GEGLU;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (geglu && (j & 1)) continue;  // gate tiles are consumed together with their value tile
      const int ncol = bn * BN + wn * 32 * NJ + j * 32 + (lane & 31);  // column in (packed) W-row space
      const bool nok = ncol < N;
      const float bv = (d.bias && nok) ? d.bias[ncol] : 0.f;
      float bg = 0.f;
      int ocol = ncol;
      if (geglu) {
        bg = (d.bias && ncol + 32 < N) ? d.bias[ncol + 32] : 0.f;
        ocol = ((ncol >> 6) << 5) + (ncol & 31);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm * BM + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m >= M || !nok) continue;
        float v = acc[i][j][r] + bv;
        if (RV) v += T::to_f32(RV[(long long)(m / d.rows_per_batch) * d.rowvec_ld + ncol]);
        if (d.act == DBIR_ACT_SILU) {
          v = silu_f(v);
        } else if (d.act == DBIR_ACT_GELU) {
          v = gelu_f(v);
        } else if (d.act == DBIR_ACT_LRELU) {
          v = v > 0.f ? v : v * d.act_param;
        } else if (geglu) {
          if constexpr (NJ >= 2) {
            const float g = acc[i][(j + 1) % NJ][r] + bg;
            v = v * gelu_f(g);
          }
        }
        v *= d.out_scale;
        if (Rg) v += T::to_f32(Rg[(long long)m * d.ldr + ocol]);
        if (d.store_mode == 0) {
          const long long off = (long long)bz * d.strideC_z + (long long)m * d.ldc + ocol;
          if (d.out_f32)
            reinterpret_cast<float*>(d.C)[off] = v;
          else
            reinterpret_cast<u16*>(d.C)[off] = T::from_f32(v);
        } else {
          const int bb = m / d.trans_L, l = m - bb * d.trans_L;
          const long long off = (long long)bb * d.trans_bstride + (long long)ocol * d.trans_ld + l;
          reinterpret_cast<u16*>(d.C)[off] = T::from_f32(v);
        }
      }
    }
  }
}

template <typename T, int MI, int NJ>
int launch(const GemmParams& p, hipStream_t s) {
  constexpr int BM = 64 * MI, BN = 64 * NJ;
  const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(u16);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, MI, NJ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(cdiv(p.d.M, BM), cdiv(p.d.N, BN), p.d.batch > 0 ? p.d.batch : 1);
  hipLaunchKernelGGL((gemm_kernel<T, MI, NJ>), grid, dim3(256), lds, s, p);
  DBIR_CHECK_LAUNCH("dbir_gemm");
  return DBIR_OK;
}

template <typename T>
int dispatch(const GemmParams& p, int tile, hipStream_t s) {
  switch (tile) {
    case 1: return launch<T, 2, 2>(p, s);
    case 2: return launch<T, 1, 2>(p, s);
    case 3: return launch<T, 1, 1>(p, s);
    case 4: return launch<T, 2, 1>(p, s);
  }
  dbir_set_error("dbir_gemm: bad tile %d", tile);
  return DBIR_ERR_ARG;
}

}  // namespace

// direct-to-LDS variant (gemm_glds.hip)
bool dbir_gemm_glds_eligible(const dbir_gemm_desc& d);
int dbir_gemm_glds(const dbir_gemm_desc& d, int Hv, int Wv, int tile, hipStream_t s);
// halo-patch 3x3 convolution kernel (gemm_halo.hip), tiles 50 / 51
bool dbir_gemm_halo_eligible(const dbir_gemm_desc& d, int tile);
int dbir_gemm_halo(const dbir_gemm_desc& d, int tile, hipStream_t s);
// persistent small-K linear kernel (gemm_pers.hip), tiles 70 - 73
bool dbir_gemm_pers_eligible(const dbir_gemm_desc& d, int tile);
int dbir_gemm_pers(const dbir_gemm_desc& d, int tile, hipStream_t s);

// fine-phase 256x320 kernel (gemm_8p.hip), tile 80
bool dbir_gemm_8p_eligible(const dbir_gemm_desc& d);
int dbir_gemm_8p(const dbir_gemm_desc& d, int Hv, int Wv, int tile, hipStream_t s);

// register-streaming linear kernel (gemm_rs.hip), tiles 93 - 97
bool dbir_gemm_rs_eligible(const dbir_gemm_desc& d, int tile);
int dbir_gemm_rs(const dbir_gemm_desc& d, int tile, hipStream_t s);
// rows per tile for which the launch that just ran emits GroupNorm column sums (dbir_gemm_desc.stats); set by the
// direct-to-LDS / halo launchers, 0 otherwise
thread_local int g_dbir_stats_rows = 0;

static int dbir_gemm_impl(const dbir_gemm_desc* dd, void* stream);

extern "C" int dbir_gemm(dbir_gemm_desc* dd, void* stream) {
  g_dbir_stats_rows = 0;
  const int rc = dbir_gemm_impl(dd, stream);
  if (dd) dd->stats_rows = rc == DBIR_OK ? g_dbir_stats_rows : 0;
  return rc;
}

static int dbir_gemm_impl(const dbir_gemm_desc* dd, void* stream) {
  DBIR_CHECK_ARG(dd && dd->A && dd->W && dd->C, "dbir_gemm: null pointer");
  GemmParams p;
  p.d = *dd;
  dbir_gemm_desc& d = p.d;
  // epilogue statistics: plain 16-bit row-major stores of the direct-to-LDS / halo kernels only
  // (split-K launches keep them since round 4: tile 80 reduces its K slices inside the launch, the other kernels' reduce
  //  pass emits them per 64-row tile)
  if (d.stats && (d.store_mode != 0 || d.out_f32 || d.act == DBIR_ACT_GEGLU || d.batch > 1 ||
                  (reinterpret_cast<uintptr_t>(d.stats) & 15)))
    d.stats = nullptr;
  DBIR_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "dbir_gemm: bad M/N/K %d %d %d", d.M, d.N, d.K);
  DBIR_CHECK_ARG(d.Kpad % BK == 0 && d.Kpad >= d.K, "dbir_gemm: Kpad %d must be a multiple of 64 and >= K %d",
                 d.Kpad, d.K);
  DBIR_CHECK_ARG(d.dtype == DBIR_F16 || d.dtype == DBIR_BF16, "dbir_gemm: bad dtype");
  if (d.mode == DBIR_MODE_LINEAR) {
    DBIR_CHECK_ARG(d.K % 8 == 0 && d.lda % 8 == 0, "dbir_gemm: linear needs K%%8==0 and lda%%8==0 (K=%d lda=%lld)",
                   d.K, d.lda);
    p.Hv = p.Wv = 0;
  } else if (d.mode == DBIR_MODE_CONV3X3) {
    if (d.upsample == 2) {   // parity-collapsed nearest-x2 upsample convolution (gemm_8p.hip PH4): W = [4][Wrows][Kpad], K = 4 Cin
      DBIR_CHECK_ARG(d.Cin % 32 == 0 && d.K == 4 * d.Cin, "dbir_gemm: upsample == 2 needs Cin%%32==0 and K==4*Cin");
      DBIR_CHECK_ARG(d.tile % 100 == 80, "dbir_gemm: upsample == 2 runs on tile 80 only");
    } else {
      DBIR_CHECK_ARG(d.Cin % 8 == 0 && d.K == 9 * d.Cin, "dbir_gemm: conv needs Cin%%8==0 and K==9*Cin");
      DBIR_CHECK_ARG(d.upsample == 0 || d.upsample == 1, "dbir_gemm: bad upsample %d", d.upsample);
    }
    DBIR_CHECK_ARG(d.stride == 1 || d.stride == 2, "dbir_gemm: conv stride must be 1 or 2");
    DBIR_CHECK_ARG((long long)d.B * d.Ho * d.Wo == d.M, "dbir_gemm: conv M != B*Ho*Wo");
    p.Hv = d.upsample == 1 ? 2 * d.Hi : d.Hi;
    p.Wv = d.upsample == 1 ? 2 * d.Wi : d.Wi;
  } else {
    dbir_set_error("dbir_gemm: bad mode %d", d.mode);
    return DBIR_ERR_ARG;
  }
  if (d.rowvec) DBIR_CHECK_ARG(d.rows_per_batch > 0, "dbir_gemm: rowvec needs rows_per_batch");
  if (d.store_mode == 1) DBIR_CHECK_ARG(d.trans_L > 0 && !d.out_f32, "dbir_gemm: bad transposed store args");
  if (d.act == DBIR_ACT_GEGLU) DBIR_CHECK_ARG(d.N % 64 == 0, "dbir_gemm: GEGLU needs packed N %% 64 == 0");
  if (d.batch <= 0) d.batch = 1;
  int tile = d.tile;
  DBIR_CHECK_ARG(tile >= 0 && tile <= 97 && tile != 13 && !(tile >= 74 && tile <= 89 && tile != 80), "dbir_gemm: bad tile %d", tile);
  if (tile >= 93) {
    DBIR_CHECK_ARG(dbir_gemm_rs_eligible(d, tile),
                   "dbir_gemm: tile %d (register-streaming linear kernel) needs a dense linear with K %% 64 == 0, M / N multiples of "
                   "the tile, a 16-bit row-major output, bias / residual epilogue only (no activation, row vector, scale, "
                   "split-K, batch, transposed or f32 store)", tile);
    d.stats = nullptr;
    return dbir_gemm_rs(d, tile, reinterpret_cast<hipStream_t>(stream));
  }
  if (tile == 80) {
    DBIR_CHECK_ARG(dbir_gemm_8p_eligible(d),
                   "dbir_gemm: tile 80 (fine-phase 256x320 kernel) needs a linear / 3x3 convolution with K / Cin %% 32 == 0, "
                   "16-byte aligned operands, a 16-bit row-major output and no GEGLU / transposed / f32 store / batch");
    return dbir_gemm_8p(d, p.Hv, p.Wv, tile, reinterpret_cast<hipStream_t>(stream));
  }
  if (tile >= 70 && tile < 80) {
    DBIR_CHECK_ARG(dbir_gemm_pers_eligible(d, tile),
                   "dbir_gemm: tile %d (persistent linear kernel) needs a dense linear with K %% 32 == 0, M a multiple of "
                   "the tile height, N %% 8 == 0, a 16-byte aligned 16-bit row-major output / residual and no row vector, "
                   "split-K, transposed or f32 store", tile);
    d.stats = nullptr;  // (the persistent kernel's register epilogue has no column-sum stage)
    return dbir_gemm_pers(d, tile, reinterpret_cast<hipStream_t>(stream));
  }
  if (tile == 0 || tile >= 5) {
    const bool ok = dbir_gemm_glds_eligible(d);
    DBIR_CHECK_ARG(ok || tile == 0, "dbir_gemm: tile %d (direct-to-LDS kernel) needs K/Cin %% 64 == 0, 16-byte aligned "
                   "operands and a 16-bit row-major output", tile);
    DBIR_CHECK_ARG(d.splitk <= 1 || ok, "dbir_gemm: split-K is implemented by the direct-to-LDS kernels only "
                   "(tiles >= 5, eligible operands)");
    if (tile >= 50 && tile < 70) {
      DBIR_CHECK_ARG(ok && d.store_mode == 0 && dbir_gemm_halo_eligible(d, tile),
                     "dbir_gemm: tile %d (halo-patch kernel) needs a stride-1 pad-1 3x3 convolution with Cin %% 64 == 0 "
                     "whose 256-row tiles are whole image rows", tile);
      return dbir_gemm_halo(d, tile, reinterpret_cast<hipStream_t>(stream));
    }
    if (ok) return dbir_gemm_glds(d, p.Hv, p.Wv, tile, reinterpret_cast<hipStream_t>(stream));
  }
  DBIR_CHECK_ARG(d.splitk <= 1, "dbir_gemm: split-K needs the direct-to-LDS kernel (tile 5-12)");
  d.stats = nullptr;  // generic register-staged kernel: no column-sum stage
  if (tile == 0) {
    // largest tile that still yields >= ~1.5 waves of blocks over the 256 CUs; GEGLU needs NJ == 2.
    const long long z = d.batch;
    auto blocks = [&](int bm_, int bn_) { return (long long)cdiv(d.M, bm_) * cdiv(d.N, bn_) * z; };
    if (blocks(128, 128) >= 384)
      tile = 1;
    else if (blocks(64, 128) >= 384 || d.act == DBIR_ACT_GEGLU)
      tile = 2;
    else
      tile = 3;
  }
  if (d.act == DBIR_ACT_GEGLU) DBIR_CHECK_ARG(tile == 1 || tile == 2, "dbir_gemm: GEGLU needs a 128-wide tile");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d.dtype == DBIR_F16 ? dispatch<F16>(p, tile, s) : dispatch<BF16>(p, tile, s);
}
