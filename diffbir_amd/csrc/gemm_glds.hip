// Implicit-GEMM (linear / 1x1 / 3x3 convolution) on gfx950 MFMA, direct-to-LDS staged variant.
//
// This is the hot kernel of the engine: every conv3x3 with Cin % 64 == 0 and every linear with K % 64 == 0
// of the UNet / ControlNet / VAE / SwinIR (i.e. all of them except the 4-/8-channel stem convs) runs here.
// The generic register-staged kernel in gemm.hip remains as the fallback for the odd shapes and for the
// f32 / transposed stores.
//
// Structure (cdna_hip_programming.md §5, "step 3" + T2):
//   * 256 threads = 4 wave64, each wave owns a 64x64 output tile = 2x2 v_mfma_f32_32x32x16 accumulators;
//     the waves are arranged WM x WN so one template gives 128x128 (2x2), 256x64 (4x1) and 64x256 (1x4)
//     block tiles.  BK = 64 halfs = one 128-byte line per tile row.
//   * Both operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (16 B per lane, no VGPR round trip).
//     The LDS image is lane-linear, so the bank-conflict swizzle is applied on the SOURCE address: LDS chunk
//     position `cpos` of row r holds logical 16-byte chunk `cpos ^ (r & 7)`; the MFMA fragment ds_read_b128
//     applies the same XOR (guide rule 21).
//   * conv3x3 is an implicit GEMM whose K order is (tap, channel); Cin % 64 == 0 makes every K tile lie inside
//     one tap, so a tile row is one contiguous 128-byte run of the NHWC input at a per-row pixel offset that
//     only changes when the tap changes.  Padding rows / taps outside the image read a 256-byte zero page.
//   * double-buffered LDS, tile t+1 in flight while tile t is multiplied; one barrier per K tile;
//     2 blocks per CU (64 KiB LDS each) interleave to cover the barrier drain.
//   * MFMA orientation is D[n][m] (first operand = weight rows) so that a lane holds 4 consecutive output
//     channels of one pixel: the f32 epilogue (bias, time-embedding row vector, SiLU/GELU/LeakyReLU/GEGLU,
//     scale) runs in registers, the 16-bit tile is transposed through LDS and written with 16-byte row-contiguous
//     stores (+ residual add on that side).
//   * blockIdx -> tile mapping is XCD-aware (bijective remap, guide T1): each XCD's L2 sees a contiguous
//     range of tiles with the N tiles of one activation panel adjacent.
#include "common.h"

namespace {

constexpr int BK = 64;

__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];  // zero-initialised device memory

struct G2Params {
  dbir_gemm_desc d;
  int Hv, Wv;   // virtual (upsampled) input extent for conv bounds checks
  int nkc;      // K tiles per tap (conv) / total K tiles (linear)
  int ntaps;    // 9 (conv) / 1 (linear)
  int mtiles, ntiles;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T, int WM, int WN>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(const G2Params p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int RA = BM / 32, RB = BN / 32;  // tile rows staged per thread
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const dbir_gemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lq = lane & 31, hi = lane >> 5;
  const int bz = blockIdx.y;

  // ---- XCD-aware tile mapping (bijective) ----
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
  const u16* zp = reinterpret_cast<const u16*>(g_zero_page);

  // ---- staging roles: thread handles LDS chunk position (row = (tid>>3) + 32*i, cpos = tid&7) ----
  const int srow = tid >> 3;
  const int cch = ((tid & 7) ^ (srow & 7)) * 8;  // logical K offset (halfs) of the chunk this thread fetches
  const bool conv = d.mode == DBIR_MODE_CONV3X3;

  int a_pix0[RA];     // conv: b*Hi*Wi ; linear: unused
  int a_yx0[RA];      // conv: (iy0 << 16) | (ix0 & 0xffff), virtual coords of tap (0,0)
  bool a_ok[RA];      // row < M
  const u16* a_rp[RA];  // current row pointer (tap applied), nullptr-equivalent = zero page when invalid
  bool a_rv[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = tm * BM + srow + 32 * i;
    a_ok[i] = m < M;
    if (conv) {
      const int hw = d.Ho * d.Wo;
      const int mm = a_ok[i] ? m : 0;
      const int b = mm / hw;
      const int rem = mm - b * hw;
      const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
      a_pix0[i] = b * d.Hi * d.Wi;
      a_yx0[i] = (oy * d.stride - d.pad) * 65536 + ((ox * d.stride - d.pad) & 0xffff);
      a_rp[i] = zp;
      a_rv[i] = false;
    } else {
      a_pix0[i] = 0;
      a_yx0[i] = 0;
      a_rv[i] = a_ok[i];
      a_rp[i] = a_ok[i] ? Ag + (long long)m * d.lda + cch : zp;
    }
  }
  const u16* w_rp[RB];
  bool w_rv[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = tn * BN + srow + 32 * i;
    w_rv[i] = n < d.Wrows;
    w_rp[i] = w_rv[i] ? Wg + (long long)n * d.Kpad + cch : zp;
  }

  // staging cursor (uniform): tile index, tap, channel-tile within the tap
  int s_kt = 0, s_tap = 0, s_cc = 0;

#define SET_TAP()                                                                                   \
  do {                                                                                              \
    if (conv) {                                                                                     \
      const int ky_ = (s_tap * 11) >> 5, kx_ = s_tap - 3 * ky_;                                     \
      _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                              \
        int iy_ = (a_yx0[i] >> 16) + ky_;                                                           \
        int ix_ = (int)(short)(a_yx0[i] & 0xffff) + kx_;                                            \
        const bool ok_ = a_ok[i] && iy_ >= 0 && iy_ < p.Hv && ix_ >= 0 && ix_ < p.Wv;               \
        if (d.upsample) {                                                                           \
          iy_ >>= 1;                                                                                \
          ix_ >>= 1;                                                                                \
        }                                                                                           \
        a_rv[i] = ok_;                                                                              \
        a_rp[i] = ok_ ? Ag + (long long)(a_pix0[i] + iy_ * d.Wi + ix_) * d.Cin + cch : zp;          \
      }                                                                                             \
    }                                                                                               \
  } while (0)

// issue the direct-to-LDS loads of K tile s_kt into buffer (buf_), then advance the cursor
#define STAGE(buf_)                                                                                 \
  do {                                                                                              \
    char* ab_ = smem + (buf_) * BUF_BYTES + wave * 1024;                                            \
    char* bb_ = ab_ + A_BYTES;                                                                      \
    const int koff_ = s_cc * BK;                                                                    \
    _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                                \
      const u16* src_ = a_rv[i] ? a_rp[i] + koff_ : zp;                                             \
      __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(ab_ + i * 4096), 16, 0, 0);           \
    }                                                                                               \
    const long long woff_ = (long long)s_kt * BK;                                                   \
    _Pragma("unroll") for (int i = 0; i < RB; ++i) {                                                \
      const u16* src_ = w_rv[i] ? w_rp[i] + woff_ : zp;                                             \
      __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(bb_ + i * 4096), 16, 0, 0);           \
    }                                                                                               \
    ++s_kt;                                                                                         \
    if (++s_cc == p.nkc) {                                                                          \
      s_cc = 0;                                                                                     \
      ++s_tap;                                                                                      \
      if (s_tap < p.ntaps) SET_TAP();                                                               \
    }                                                                                               \
  } while (0)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.nkc * p.ntaps;
  SET_TAP();
  STAGE(0);

  // fragment read offsets (bytes) inside a tile: row * 128 + ((2*ks + hi) ^ (row & 7)) * 16, row & 7 == lq & 7
  const int a_frag = (wm * 64 + lq) * 128;
  const int b_frag = A_BYTES + (wn * 64 + lq) * 128;
  const int sw = lq & 7;

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt has landed for every wave; everyone is done reading buffer cur^1
    if (kt + 1 < nk) STAGE(cur ^ 1);
    const char* base = smem + cur * BUF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int co = ((2 * ks + hi) ^ sw) * 16;
      typename T::vec8 xf[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        xf[i] = *reinterpret_cast<const typename T::vec8*>(base + a_frag + i * 4096 + co);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wf[j] = *reinterpret_cast<const typename T::vec8*>(base + b_frag + j * 4096 + co);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = T::mfma32(wf[j], xf[i], acc[i][j]);  // D[n][m]
    }
  }
#undef STAGE
#undef SET_TAP

  // ---------------- epilogue: f32 math in registers -> 16-bit tile in LDS -> row-contiguous 16 B stores ----------
  const bool geglu = d.act == DBIR_ACT_GEGLU;
  const int N = d.N;
  const int bn_out = geglu ? BN / 2 : BN;
  const int cs_ld = bn_out + 8;  // halfs; (bn_out + 8) * 2 B is a multiple of 16
  u16* Cs = reinterpret_cast<u16*>(smem);
  const u16* __restrict__ RV = reinterpret_cast<const u16*>(d.rowvec);
  __syncthreads();  // all waves finished reading the operand tiles
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + lq;
    const int m = tm * BM + row;
    const int mb = (m < M ? m : M - 1);
    const u16* rvp = RV ? RV + (long long)(mb / d.rows_per_batch) * d.rowvec_ld : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (geglu && j == 1) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = wn * 64 + j * 32 + 8 * g + 4 * hi;  // local packed column of element 0
        const int n0 = tn * BN + nl;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + e;
          float x = acc[i][j][4 * g + e];
          if (d.bias && n < N) x += d.bias[n];
          if (rvp && n < N) x += T::to_f32(rvp[n]);
          if (d.act == DBIR_ACT_SILU) {
            x = silu_f(x);
          } else if (d.act == DBIR_ACT_GELU) {
            x = gelu_f(x);
          } else if (d.act == DBIR_ACT_LRELU) {
            x = x > 0.f ? x : x * d.act_param;
          } else if (geglu) {
            float gte = acc[i][1][4 * g + e];
            if (d.bias && n + 32 < N) gte += d.bias[n + 32];
            x = x * gelu_f(gte);
          }
          v[e] = x * d.out_scale;
        }
        const int ocl = geglu ? (wn * 32 + 8 * g + 4 * hi) : nl;
        uint2 pk;
        pk.x = (uint32_t)T::from_f32(v[0]) | ((uint32_t)T::from_f32(v[1]) << 16);
        pk.y = (uint32_t)T::from_f32(v[2]) | ((uint32_t)T::from_f32(v[3]) << 16);
        *reinterpret_cast<uint2*>(Cs + row * cs_ld + ocl) = pk;
      }
    }
  }
  __syncthreads();
  {
    const int n_out = geglu ? N / 2 : N;
    const int ch_per_row = bn_out >> 3;
    const int total = BM * ch_per_row;
    const u16* __restrict__ Rg = d.R ? reinterpret_cast<const u16*>(d.R) + (long long)bz * d.strideR_z : nullptr;
    u16* __restrict__ Cg = reinterpret_cast<u16*>(d.C) + (long long)bz * d.strideC_z;
    for (int q = tid; q < total; q += 256) {
      const int row = q / ch_per_row, ch = q - row * ch_per_row;
      const int m = tm * BM + row;
      const int ncol = tn * bn_out + ch * 8;
      if (m >= M || ncol >= n_out) continue;
      uint4 v = *reinterpret_cast<const uint4*>(Cs + row * cs_ld + ch * 8);
      if (ncol + 8 <= n_out) {
        if (Rg) {
          const uint4 rr = *reinterpret_cast<const uint4*>(Rg + (long long)m * d.ldr + ncol);
          float a[8], b[8];
          unpack8<T>(v, a);
          unpack8<T>(rr, b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
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

template <typename T, int WM, int WN>
int launch2(G2Params& p, hipStream_t s) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int lds = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<T, WM, WN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  p.mtiles = cdiv(p.d.M, BM);
  p.ntiles = cdiv(p.d.N, BN);
  dim3 grid((unsigned)(p.mtiles * p.ntiles), p.d.batch > 0 ? p.d.batch : 1);
  hipLaunchKernelGGL((gemm_glds_kernel<T, WM, WN>), grid, dim3(256), lds, s, p);
  DBIR_CHECK_LAUNCH("dbir_gemm(glds)");
  return DBIR_OK;
}

template <typename T>
int dispatch2(G2Params& p, int tile, hipStream_t s) {
  switch (tile) {
    case 5: return launch2<T, 2, 2>(p, s);
    case 6: return launch2<T, 4, 1>(p, s);
    case 7: return launch2<T, 1, 4>(p, s);
  }
  dbir_set_error("dbir_gemm: bad glds tile %d", tile);
  return DBIR_ERR_ARG;
}

}  // namespace

// Is the descriptor (already validated by dbir_gemm) runnable on the direct-to-LDS kernel?
bool dbir_gemm_glds_eligible(const dbir_gemm_desc& d) {
  if (d.store_mode != 0 || d.out_f32) return false;
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
  if (dd.mode == DBIR_MODE_LINEAR) {
    p.nkc = dd.K / BK;
    p.ntaps = 1;
  } else {
    p.nkc = dd.Cin / BK;
    p.ntaps = 9;
  }
  if (tile == 0) {
    // N = 320 (the 64x64-latent level of the UNet) tiles exactly with 64-wide column tiles; otherwise 128x128.
    const int waste128 = cdiv(dd.N, 128) * 128 - dd.N, waste64 = cdiv(dd.N, 64) * 64 - dd.N;
    tile = (dd.act != DBIR_ACT_GEGLU && waste64 < waste128 && dd.M >= 2048) ? 6 : 5;
  }
  return dd.dtype == DBIR_F16 ? dispatch2<F16>(p, tile, s) : dispatch2<BF16>(p, tile, s);
}
