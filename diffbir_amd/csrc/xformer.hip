// Fused transformer-block kernels for the 64x64 latent level (C = 320) of the SD-2.1 UNet / ControlNet
// (reference diffbir/model/attention.py:19-45 GEGLU / FeedForward, 189-216 CrossAttention, 265-274
// BasicTransformerBlock._forward, 334-353 SpatialTransformer.forward).
//
// Why: at C = 320 every linear of the block is a small-K GEMM that moves 84 - 250 MB through HBM for 13 - 54 GFLOP
// and runs at 385 - 740 TF/s as its own launch (profiles/r2_autotune_persistent_call11.log), and the LayerNorms /
// text cross-attention between them are pure HBM round trips: one block moves ~1.85 GB.  But everything in the block
// except the self-attention is ROW-LOCAL: a 128-row panel of the [B*4096, 320] activation (80 KB as f16) fits in LDS,
// so the whole chain can run on-chip per panel with only the weights streaming in (from L2, they are shared by all
// panels) — two kernels replace 16 launches:
//
//   xf_head  : GroupNorm-apply(x) -> proj_in -> h ; LayerNorm1(h) -> q | k | v^T          (before the self-attention)
//   xf_tail  : h1 = attn @ Wo1 + h ; q = LN2(h1) @ Wq ; a = softmax(q K_ctx^T) V_ctx (text context, Lk <= 96, all
//              heads) ; h2 = a @ Wo2 + h1 ; h3 = GEGLU-FF(LN3(h2)) + h2 ; out = h3 @ Wpo + x   (after it)
//
// Structure (one workgroup = 512 threads = 8 wave64 per CU, persistent over panels, XCD-contiguous panel ranges):
//   * the panel lives in LDS in MFMA-fragment-major order X[rowblk 4][kstep 20][lane 64][16 B]: the A-side fragment
//     of (32 rows, 16 k) is one contiguous KB (conflict-free ds_read_b128), and the accumulator layout D[n][m] (lane =
//     row, 4 consecutive columns per register quad) writes the NEXT GEMM's operand back in place with ds_write_b64;
//   * a wave owns 32 rows x 160 columns (WM 4 x WN 2, 5 accumulator blocks = 80 VGPRs): a full N = 320 output row
//     panel is register-resident, so a GEMM output never leaves the CU before LayerNorm / the next GEMM consume it;
//   * the residual stream rides in registers: packed 16-bit (40 VGPRs) between GEMMs and as the INITIAL VALUE of the
//     f32 accumulators of the GEMM that adds to it (h + b + a W^T rounded once; the reference rounds the GEMM output and
//     the sum separately — the fused form is the more accurate one);
//   * ALL weights of the block are packed on the host (diffbir_amd/xformer.py) into ONE stream of 20.5 KB tiles in
//     consumption order, each tile already in LDS image order (20 fragment pieces of 1 KB = [32 rows][16 k] + 512 B of
//     f32 side data: the GEGLU bias of the chunk), so staging is a linear `buffer_load ... lds` copy through a 3-slot
//     ring with counted vmcnt, running ahead across phase boundaries (the weights do not depend on computed data);
//   * GEGLU feed-forward in 20 chunks of 64 hidden units: G = n3 W1_c^T (+b, from the tile's side data) -> x*gelu(gate)
//     -> 16 KB LDS chunk -> out += g W2_c^T: the [M, 1280] hidden tensor (168 MB write + read per call) never exists;
//   * text cross-attention per (32-row block, head) unit on one wave: K / V^T fragments come pre-arranged per (batch,
//     head) from the context cache (fragment order, 16-byte coalesced loads, L2-resident), S^T = K q^T, one-pass
//     softmax, O^T = V^T P^T with P fed from the accumulator registers (same dataflow as attention.hip).
//
// vmcnt discipline (gfx9 counts loads and stores in one counter, loads return in order): ring waits are counted with
// N = the tile loads this wave issued after the awaited tile; any other outstanding VMEM op only makes such a wait
// stricter, never wrong.  Every phase epilogue ends with an explicit vmcnt(0) after which `landed = issued`.
#include <stdint.h>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

// Per-width configuration.  C = 320 (64x64 level): WM 4 x WN 2 waves over a 128-row panel; C = 640 (32x32 level): WM 2 x
// WN 4 over a 64-row panel (the panel image is 80 KB either way) — a wave owns 32 rows x 160 columns in both.  A weight tile
// is always 20 pieces of 1 KB: [GNKS k-steps][C / 32 column blocks] for the N = C GEMMs and the feed-forward output
// projection, [5 k-steps][4 blocks = (value, gate) x 2 hidden blocks] for the GEGLU projection; at WN = 4 a feed-forward
// chunk (32 WN hidden units) takes NSUB = 2 such tile runs, each multiplied by the half of the waves that owns its blocks.
template <int CC>
struct XfCfg {
  static_assert(CC == 320 || CC == 640, "fused transformer kernels: C = 320 or 640");
  static constexpr int C = CC, WN = CC / 160, WM = 8 / WN, BM = 32 * WM, KS = CC / 16, HID = 4 * CC, HEADS = CC / 64;
  static constexpr int NB = CC / 32;              // column blocks of an N = C GEMM
  static constexpr int GNKS = 20 / NB;            // k-steps per tile of an N = C GEMM
  static constexpr int GNT = KS / GNKS;           // tiles of a C x C GEMM
  static constexpr int NSUB = WN / 2;             // GEGLU projection runs per feed-forward chunk
  static constexpr int CHH = 32 * WN;             // hidden units per chunk
  static constexpr int CH = HID / CHH;            // chunks (20)
  static constexpr int F1T = KS / 5;              // tiles per GEGLU projection run
  static constexpr int GKST = CHH / 16;           // k-steps of the chunk image
  static constexpr int F2T = GKST / GNKS;         // tiles of a chunk's output projection
  static constexpr int TAIL_TILES = 4 * GNT + CH * (NSUB * F1T + F2T);  // out1, q2, out2, CH x (FF1 + FF2), proj_out
  static constexpr int HEAD_TILES = 4 * GNT;                            // proj_in, q, k, v
};
constexpr int XNT = 512;                // threads per workgroup
constexpr int TILE_W = 20480, TILE_AUX = 512, TILE_BYTES = TILE_W + TILE_AUX;
constexpr int X_BYTES = 128 * 320 * 2;  // 81920 = BM * C * 2 for both configurations
constexpr int NSLOT = 3;
constexpr int RING_OFF = X_BYTES;
constexpr int GB_OFF = RING_OFF + NSLOT * TILE_BYTES;  // 144896: GEGLU chunk [WM][GKST][64][16 B] / LayerNorm partials
constexpr int GB_BYTES = 16384;
constexpr int XF_LDS = GB_OFF + GB_BYTES;              // 161280 <= 163840
constexpr int XKB = 3, XLKP = 96;       // padded text context: 3 key blocks of 32
// the configuration's constants under the names the macros below use (declared at the top of each kernel)
#define XF_CFG(CC)                                                                                                      \
  using G_ = XfCfg<CC>;                                                                                                 \
  constexpr int XC = G_::C, XBM = G_::BM, XKS = G_::KS, XCH = G_::CH, XHEADS = G_::HEADS, XWM = G_::WM, XWN = G_::WN,   \
                XNB = G_::NB, GNKS = G_::GNKS, GNT = G_::GNT, NSUB = G_::NSUB, F1T = G_::F1T, GKST = G_::GKST,          \
                F2T = G_::F2T, TAIL_TILES = G_::TAIL_TILES, HEAD_TILES = G_::HEAD_TILES;                                \
  (void)XCH; (void)XHEADS; (void)NSUB; (void)F1T; (void)GKST; (void)F2T; (void)TAIL_TILES; (void)HEAD_TILES;           \
  (void)XWM; (void)GNT; (void)XNB; (void)GNKS

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// every global access goes through a buffer descriptor: one shared 32-bit per-lane offset VGPR + a scalar offset + an
// immediate — nothing per-access for the compiler to hoist into 64-bit per-lane pointers (which spilled the kernel)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xsrd(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7ffffe00LL ? 0x7ffffe00LL : bytes), 0x00020000);
}
__device__ __forceinline__ float4 xld_f4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ uint4 xld_u4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// 16-byte stores take NO scalar offset: with an SGPR soffset hipcc's hazard recognizer assumes a buffer store has read its
// data registers at issue and lets a VALU write of them follow immediately — on gfx950 the last lanes of each 16-lane
// group then store the NEW value (measured: wrong h rows, lanes 12-15 / 28-31, non-reproducible).  With soffset = 0 the
// compiler inserts the wait states itself (as it does for global_store_dwordx4).
__device__ __forceinline__ void xst_u4(__amdgpu_buffer_rsrc_t r, int voff, uint4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}
__device__ __forceinline__ void xst_u2(__amdgpu_buffer_rsrc_t r, int voff, int soff, uint2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, 0);
}

template <int N>
__device__ __forceinline__ void xwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void xbarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct XfParams {
  // activations (16-bit, row-major, row strides in elements)
  const u16* o;  long long ldo;    // tail: self-attention output [Msrc, C]        head: block input x [M, C]
  const u16* h;  long long ldh;    // tail: residual stream before out1 [Msrc, C]  head: (out) h = proj_in(GN(x)) [M, C]
  const u16* x;  long long ldx;    // tail: block input (residual of proj_out) [Msrc, C]
  u16* out;      long long ldout;  // tail: block output [M, C]                     head: (out) q|k [M, 2C]
  u16* vt;       long long vt_ld, vt_bs;  // head: (out) v^T [B, C, Lpad]
  const float* ab;                 // head: GroupNorm scale | shift per (sample, channel) f32 [B, 2, C]
  int M, L;                        // output rows, rows per sample (L % 128 == 0)
  int pair_bs;                     // tail: 0, or bs: inputs hold the DISTINCT samples of a CFG batch [G*bs], outputs the
                                   // full batch [G*2*bs]: source sample of b = (b / (2 bs)) * bs + b % bs
  const void* wstream;             // packed weight tile stream (TAIL_TILES / HEAD_TILES tiles of TILE_BYTES)
  const float* prm;                // f32 parameter rows of C floats (see xformer.py)
  const u16* kf; const u16* vf;    // tail: context K / V^T fragments [B][heads][3][4][64][8], [B][heads][2][6][64][8]
  int Lk;  float c;                // context length, softmax scale * log2(e)
  int npanels, q, gx;              // panels, panels / workgroups per XCD
  int stop_after;                  // DEBUG (tests): dump an intermediate instead of the result, see dbir.h
  int variant;                     // A/B (dbir_set_option DBIR_OPT_XF_VARIANT): 0 = every wave stages before its MFMAs
  int prm_row_bytes;               // = C * 4, as a RUNTIME value: a parameter row's offset then stays in the scalar offset
                                   // operand (a literal is folded into per-access offset VGPRs, which the compiler hoisted
                                   // out of the panel loop and spilled: scratch reload -> wait -> load chains, 15 us per LN)
};

// ---------------------------------------------------------------------------------------------------------------
// shared machinery (macros: everything stays in registers / scalar state of the enclosing kernel)
// ---------------------------------------------------------------------------------------------------------------
// issue the direct-to-LDS copy of stream tile s_t into ring slot s_slot: 2 full 8 KB passes by all 8 waves, the
// last 4 KB by waves 0-3, the 512 B of side data by the low half of wave 4 -> 3 loads for waves 0-4, 2 for waves 5-7
#define XF_STAGE(NTILES)                                                                                         \
  do {                                                                                                            \
    char* dst_ = smem + RING_OFF + s_slot * TILE_BYTES;                                                           \
    const int so_ = s_t * TILE_BYTES;                                                                             \
    if (!(abl & 8)) {                                                                                             \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (lptr_t)(dst_ + wave * 1024), 16, w_voff, so_, 0, 0);         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (lptr_t)(dst_ + 8192 + wave * 1024), 16, w_voff, so_ + 8192,  \
                                             0, 0);                                                               \
    if (wave < 4) {                                                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (lptr_t)(dst_ + 16384 + wave * 1024), 16, w_voff,           \
                                               so_ + 16384, 0, 0);                                                \
    } else if (wave == 4) {                                                                                       \
      if (lane < 32)                                                                                              \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (lptr_t)(dst_ + TILE_W), 16, lane16, so_ + TILE_W, 0, 0); \
    }                                                                                                             \
    }                                                                                                             \
    ++issued;                                                                                                     \
    s_slot = (s_slot + 1 == NSLOT) ? 0 : s_slot + 1;                                                              \
    s_t = (s_t + 1 == (NTILES)) ? 0 : s_t + 1;                                                                    \
  } while (0)

// wait until the next tile of the stream has landed (this wave's share), publish with the workgroup barrier (which
// also orders every earlier LDS write of the workgroup), and point `tbase` at it
#define XF_ACQUIRE()                                                                                              \
  do {                                                                                                            \
    if (consumed >= landed) {                                                                                     \
      const int after_ = issued - consumed - 1;                                                                   \
      if (after_ >= 2) {                                                                                          \
        if (wave <= 4) xwait<6>(); else xwait<4>();                                                               \
      } else if (after_ == 1) {                                                                                   \
        if (wave <= 4) xwait<3>(); else xwait<2>();                                                               \
      } else {                                                                                                    \
        xwait<0>();                                                                                               \
      }                                                                                                           \
    }                                                                                                             \
    xbarrier();                                                                                                   \
    tbase = smem + RING_OFF + c_slot * TILE_BYTES;                                                                \
    c_slot = (c_slot + 1 == NSLOT) ? 0 : c_slot + 1;                                                              \
    ++consumed;                                                                                                   \
  } while (0)

// A RUN of tiles = consecutive stream tiles multiplied against one operand image (a whole N = 320 GEMM, the 4 GEGLU
// projection tiles or the 2 output tiles of a feed-forward chunk).  Fragment reads are software-pipelined one k-step
// ahead ACROSS tile boundaries: the acquire of tile i + 1 (counted wait + workgroup barrier) sits in front of the LAST
// k-step's MFMAs of tile i, when every fragment of tile i is already in registers, and is followed at once by the reads
// of tile i + 1's first fragments — so the LDS latency and the barrier skew of a tile boundary hide under 5 (or 2) armed
// MFMAs instead of standing between two tiles (measured before: 1300 - 1600 cycles per tile for 640 cycles of matrix
// work per SIMD with the barrier at the tile boundary).  Ring invariant unchanged: at the acquire of tile t every wave
// has completed its reads of tile t - 1 (lgkmcnt(0) in front of the barrier), so that slot is refilled right behind it.
//   XF_RUN_BEGIN: acquire the run's first tile, refill, read its first fragments (set 0)
//   XF_RUN_BODY : NT tiles of NKS k-steps; A-side fragment of (tile i, k-step ks) at AADDR(i, ks) (lane offset folded
//                 in), W fragments of block jl at tbase + (ks * PSTR + WFIRST + jl) * 1024 + lane * 16
#define XF_READ_FRAGS(SET, NJ, AADDR_, WPIECE)                                                                    \
  if (act_ && !(abl & 32)) {                                                                                              \
    xfr[SET] = *reinterpret_cast<const typename T::vec8*>(AADDR_);                                                \
    _Pragma("unroll") for (int jl = 0; jl < NJ; ++jl) wfr[SET][jl] =                                              \
        *reinterpret_cast<const typename T::vec8*>(tbase + ((WPIECE) + jl) * 1024 + lane16);                      \
  }

#define XF_RUN_ACQ(NJ, AADDR, WFIRST)                                                                             \
  do {                                                                                                            \
    XF_ACQUIRE();                                                                                                 \
    XF_READ_FRAGS(0, NJ, AADDR(0, 0), WFIRST);                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
  } while (0)
#define XF_RUN_BEGIN(NJ, AADDR, WFIRST, NTILES)                                                                   \
  do {                                                                                                            \
    XF_RUN_ACQ(NJ, AADDR, WFIRST);                                                                                \
    if (issued < total) XF_STAGE(NTILES);                                                                         \
  } while (0)

#define XF_RUN_BODY(NT, NKS, NJ, PSTR, AADDR, WFIRST, ACC, NTILES)                                                \
  do {                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NT; ++i) _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {           \
      const int cur_ = (i * NKS + ks) & 1;                                                                        \
      if (ks + 1 < NKS) {                                                                                         \
        XF_READ_FRAGS(cur_ ^ 1, NJ, AADDR(i, ks + 1), (ks + 1) * (PSTR) + (WFIRST));                              \
      } else if (i + 1 < NT) {                                                                                    \
        XF_ACQUIRE();                                                                                             \
        XF_READ_FRAGS(cur_ ^ 1, NJ, AADDR(i + 1, 0), WFIRST);                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (issued < total) XF_STAGE(NTILES);                                                                     \
      }                                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                          \
      if (act_ && !(abl & 16)) {                                                                                  \
        _Pragma("unroll") for (int jl = 0; jl < NJ; ++jl) ACC[jl] = T::mfma32(wfr[cur_][jl], xfr[cur_], ACC[jl]); \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)

// a whole N = C, K = C GEMM over the panel in X: GNT tiles of (C / 32 blocks x GNKS k-steps)
#define XF_A_XG(i, ks) (smem + (wm * XKS + GNKS * (i) + (ks)) * 1024 + lane16)
#define XF_A_X5(i, ks) (smem + (wm * XKS + 5 * (i) + (ks)) * 1024 + lane16)
#define XF_A_GB(i, ks) (smem + GB_OFF + (wm * GKST + GNKS * (i) + (ks)) * 1024 + lane16)
#define XF_GEMMCC(NTILES)                                                                                         \
  do {                                                                                                            \
    XF_RUN_BEGIN(5, XF_A_XG, 5 * wn, NTILES);                                                                     \
    XF_RUN_BODY(GNT, GNKS, 5, XNB, XF_A_XG, 5 * wn, acc, NTILES);                                                 \
  } while (0)

// byte offset inside a fragment-major operand image with KST k-steps per 32-row block of the 4 consecutive columns
// [col, col + 4) (col % 4 == 0) of row (rowblk, lq): piece (rowblk * KST + col / 16), 16-byte slot of lane
// ((col / 8) & 1) * 32 + lq, 8-byte half (col / 4) & 1
__device__ __forceinline__ int xoff(int rowblk, int kst, int col, int lq) {
  return ((rowblk * kst + (col >> 4)) * 2 + ((col >> 3) & 1)) * 512 + lq * 16 + ((col >> 2) & 1) * 8;
}

// acc (f32, D[n][m] layout) -> 16-bit operand image (X or the GEGLU chunk) — 20 / 4 ds_write_b64 per lane
// (the quad of columns 160 wn + 32 j + 8 g + 4 hi sits at xoff(wm, XKS, 160 wn + 4 hi, lq) + (2 j + g / 2) * 1024 +
// (g % 2) * 512: one opaque lane base + literals that fold into the ds_write offset field)
#define XF_STORE_X(ACC)                                                                                           \
  do {                                                                                                            \
    int xb_ = xoff(wm, XKS, 160 * wn + 4 * hi, lq);                                                               \
    XF_OPAQUE(xb_);                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int g = 0; g < 4; ++g) {                 \
      uint2 pk_;                                                                                                  \
      pk_.x = T::pack2(ACC[j][4 * g + 0], ACC[j][4 * g + 1]);                                                     \
      pk_.y = T::pack2(ACC[j][4 * g + 2], ACC[j][4 * g + 3]);                                                     \
      *reinterpret_cast<uint2*>(smem + xb_ + (2 * j + (g >> 1)) * 1024 + (g & 1) * 512) = pk_;                    \
    }                                                                                                             \
  } while (0)

// A lane-offset VGPR made opaque at its use site: "base + literal" offsets are loop-invariant, LICM hoists every one of
// them out of the panel loop into its own VGPR (40 - 60 of them, spilled, reloaded through scratch in front of each load:
// measured 15 us per LayerNorm); behind this no-op the literal stays next to the load and folds into its `offset:` field.
#define XF_OPAQUE(V) asm volatile("" : "+v"(V))

// accumulators := f32 row of C parameters (bias), columns of this lane
#define XF_ACC_BIAS(PROW)                                                                                         \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int g = 0; g < 4; ++g) {                 \
      const float4 b_ = xld_f4(prm_srd, prm_voff + (32 * j + 8 * g) * 4, (PROW) * p.prm_row_bytes);                      \
      acc[j][4 * g + 0] = b_.x;                                                                                   \
      acc[j][4 * g + 1] = b_.y;                                                                                   \
      acc[j][4 * g + 2] = b_.z;                                                                                   \
      acc[j][4 * g + 3] = b_.w;                                                                                   \
    }                                                                                                             \
  } while (0)

// accumulators += the residual stream held packed in hres (same lane layout)
#define XF_ACC_ADD_HRES()                                                                                         \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int g = 0; g < 4; ++g) {                 \
      acc[j][4 * g + 0] += T::to_f32((u16)(hres[j][g].x & 0xffff));                                               \
      acc[j][4 * g + 1] += T::to_f32((u16)(hres[j][g].x >> 16));                                                  \
      acc[j][4 * g + 2] += T::to_f32((u16)(hres[j][g].y & 0xffff));                                               \
      acc[j][4 * g + 3] += T::to_f32((u16)(hres[j][g].y >> 16));                                                  \
    }                                                                                                             \
  } while (0)

// 16-bit residual rows from HBM: after the half-wave exchange a lane owns 8 consecutive columns (one 16-byte access,
// gemm_pers.hip); RES[j][gp] = columns [160 wn + 32 j + 8 (2 gp + hi), +8) of row (32 wm + lq)
#define XF_RES_LOAD(RES, SRD, LD, ROW0)                                                                           \
  do {                                                                                                            \
    int vo_ = (int)(((32 * wm + lq) * (LD) + 160 * wn + 8 * hi) * 2);                                             \
    XF_OPAQUE(vo_);                                                                                               \
    const int so_ = (int)((long long)(ROW0) * (LD) * 2);                                                          \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int gp = 0; gp < 2; ++gp)                \
        RES[j][gp] = xld_u4(SRD, vo_ + (32 * j + 16 * gp) * 2, so_);                                              \
  } while (0)
// accumulators += RES (undo the exchange: (pk[2gp].x, pk[2gp+1].x) = swap(u.x, u.z), (.y) = swap(u.y, u.w))
#define XF_ACC_ADD_RES(RES)                                                                                       \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {              \
      const auto sx_ = __builtin_amdgcn_permlane32_swap(RES[j][gp].x, RES[j][gp].z, false, false);                \
      const auto sy_ = __builtin_amdgcn_permlane32_swap(RES[j][gp].y, RES[j][gp].w, false, false);                \
      _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                          \
        const int g = 2 * gp + q_;                                                                                \
        acc[j][4 * g + 0] += T::to_f32((u16)(sx_[q_] & 0xffff));                                                  \
        acc[j][4 * g + 1] += T::to_f32((u16)(sx_[q_] >> 16));                                                     \
        acc[j][4 * g + 2] += T::to_f32((u16)(sy_[q_] & 0xffff));                                                  \
        acc[j][4 * g + 3] += T::to_f32((u16)(sy_[q_] >> 16));                                                     \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
// round the accumulators to 16 bit and store rows with 16-byte stores (half-wave exchange as in gemm_pers.hip)
#define XF_ROW_STORE(SRD, LD, ROW0, COL0)                                                                         \
  do {                                                                                                            \
    int vo_ = (int)(((32 * wm + lq) * (LD) + 160 * wn + 8 * hi) * 2) +                                            \
              (int)(((long long)(ROW0) * (LD) + (COL0)) * 2);                                                     \
    XF_OPAQUE(vo_);                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int gp = 0; gp < 2; ++gp) {              \
      const uint32_t ax_ = T::pack2(acc[j][8 * gp + 0], acc[j][8 * gp + 1]);                                      \
      const uint32_t ay_ = T::pack2(acc[j][8 * gp + 2], acc[j][8 * gp + 3]);                                      \
      const uint32_t bx_ = T::pack2(acc[j][8 * gp + 4], acc[j][8 * gp + 5]);                                      \
      const uint32_t by_ = T::pack2(acc[j][8 * gp + 6], acc[j][8 * gp + 7]);                                      \
      const auto sx_ = __builtin_amdgcn_permlane32_swap(ax_, bx_, false, false);                                  \
      const auto sy_ = __builtin_amdgcn_permlane32_swap(ay_, by_, false, false);                                  \
      xst_u4(SRD, vo_ + (32 * j + 16 * gp) * 2, make_uint4(sx_[0], sy_[0], sx_[1], sy_[1]));                      \
    }                                                                                                             \
  } while (0)

// residual stream: round the accumulators to 16 bit -> hres (packed) and back into acc as the rounded f32 values
#define XF_ROUND_TO_HRES()                                                                                        \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int g = 0; g < 4; ++g) {                 \
      hres[j][g].x = T::pack2(acc[j][4 * g + 0], acc[j][4 * g + 1]);                                              \
      hres[j][g].y = T::pack2(acc[j][4 * g + 2], acc[j][4 * g + 3]);                                              \
      acc[j][4 * g + 0] = T::to_f32((u16)(hres[j][g].x & 0xffff));                                                \
      acc[j][4 * g + 1] = T::to_f32((u16)(hres[j][g].x >> 16));                                                   \
      acc[j][4 * g + 2] = T::to_f32((u16)(hres[j][g].y & 0xffff));                                                \
      acc[j][4 * g + 3] = T::to_f32((u16)(hres[j][g].y >> 16));                                                   \
    }                                                                                                             \
  } while (0)

// LayerNorm over the 320 columns of every panel row, input = acc (f32 values), normalised rows (16-bit) written into X.
// NO affine map here: gamma is folded into the consuming GEMM's weights and beta into its bias on the host
// (W diag(gamma), W beta: diffbir_amd/xformer.py) — the 40 parameter loads per lane were ~5 us of exposed L2 latency.
// A row is spread over 80 registers of a lane, its partner lane (lane ^ 32) and the partner wave (wn ^ 1): two-pass
// statistics (mean, then centred sum of squares) with one LDS exchange each.  The two barriers also order the
// preceding K loop's last reads of X before the writes below.
#define XF_LAYERNORM_TO_X()                                                                                       \
  do {                                                                                                            \
    float* red_ = reinterpret_cast<float*>(smem + GB_OFF);                                                        \
    const int row_ = 32 * wm + lq;                                                                                \
    float s_ = 0.f;                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int r = 0; r < 16; ++r) s_ += acc[j][r]; \
    s_ += __shfl_xor(s_, 32, 64);                                                                                 \
    if (hi == 0) red_[wn * XBM + row_] = s_;                                                                      \
    xbarrier();                                                                                                   \
    float mean_ = 0.f;                                                                                            \
    _Pragma("unroll") for (int w_ = 0; w_ < XWN; ++w_) mean_ += red_[w_ * XBM + row_];                            \
    mean_ *= (1.0f / XC);                                                                                         \
    float q_ = 0.f;                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int r = 0; r < 16; ++r) {                \
      acc[j][r] -= mean_;                                                                                         \
      q_ += acc[j][r] * acc[j][r];                                                                                \
    }                                                                                                             \
    q_ += __shfl_xor(q_, 32, 64);                                                                                 \
    if (hi == 0) red_[(XWN + wn) * XBM + row_] = q_;                                                              \
    xbarrier();                                                                                                   \
    float var_ = 0.f;                                                                                             \
    _Pragma("unroll") for (int w_ = 0; w_ < XWN; ++w_) var_ += red_[(XWN + w_) * XBM + row_];                     \
    const float rstd_ = rsqrtf(var_ * (1.0f / XC) + 1e-5f);                                                       \
    int xb_ = xoff(wm, XKS, 160 * wn + 4 * hi, lq);                                                               \
    XF_OPAQUE(xb_);                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int g = 0; g < 4; ++g) {                 \
      uint2 pk_;                                                                                                  \
      pk_.x = T::pack2(acc[j][4 * g + 0] * rstd_, acc[j][4 * g + 1] * rstd_);                                     \
      pk_.y = T::pack2(acc[j][4 * g + 2] * rstd_, acc[j][4 * g + 3] * rstd_);                                     \
      *reinterpret_cast<uint2*>(smem + xb_ + (2 * j + (g >> 1)) * 1024 + (g & 1) * 512) = pk_;                    \
    }                                                                                                             \
  } while (0)

// DEBUG dumps (tests): the operand image X, or the packed residual stream, as row-major [128, 320] rows of `out`
#define XF_DUMP_X()                                                                                               \
  do {                                                                                                            \
    xbarrier();                                                                                                   \
    for (int q_ = tid; q_ < XBM * XC / 8; q_ += XNT) {                                                            \
      const int piece_ = q_ >> 6, l_ = q_ & 63;                                                                   \
      const int rb_ = piece_ / XKS, ks_ = piece_ - rb_ * XKS;                                                     \
      const uint4 v_ = *reinterpret_cast<const uint4*>(smem + q_ * 16);                                           \
      xst_u4(out_srd, (int)(((row0 + rb_ * 32 + (l_ & 31)) * p.ldout + ks_ * 16 + (l_ >> 5) * 8) * 2), v_);       \
    }                                                                                                             \
    xbarrier();                                                                                                   \
  } while (0)
#define XF_DUMP_HRES()                                                                                            \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) _Pragma("unroll") for (int g = 0; g < 4; ++g)                   \
        xst_u2(out_srd, (int)(((32 * wm + lq) * p.ldout + 160 * wn + 32 * j + 8 * g + 4 * hi) * 2),               \
               (int)(row0 * p.ldout * 2), hres[j][g]);                                                            \
  } while (0)

// DEBUG: after a dump, consume the rest of this panel's tiles (the stream position must stay aligned with the panels)
#define XF_SKIP_REST(NTILES)                                                                                      \
  while (consumed < (pi + 1) * (NTILES)) {                                                                        \
    XF_ACQUIRE();                                                                                                 \
    if (issued < total) XF_STAGE(NTILES);                                                                         \
  }

// panel (128 rows x 320 columns, row stride LD) -> X by direct-to-LDS loads: pass pp of wave w fills piece 8 pp + w
#define XF_LOAD_PANEL(SRD, LD, ROW0)                                                                              \
  do {                                                                                                            \
    _Pragma("unroll") for (int pp = 0; pp < 10; ++pp) {                                                           \
      const int piece_ = 8 * pp + wave;                                                                           \
      const int rb_ = piece_ / XKS, ks_ = piece_ - rb_ * XKS;                                                     \
      const int so_ = (int)((((long long)(ROW0) + rb_ * 32) * (LD) + ks_ * 16) * 2);                              \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(SRD, (lptr_t)(smem + piece_ * 1024), 16, pan_voff, so_, 0, 0);     \
    }                                                                                                             \
  } while (0)

// DEBUG anatomy (stop_after == 99, debug instantiation only): s_memtime per section, summed over this workgroup's panels,
// written by wave 0 as 16 x u64 per workgroup into `out` (tools/probes/xf_anatomy.py)
#define XF_TS(I)                                                        \
  do {                                                                  \
    if (DBG == 2) {                                                     \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
      ta[I] += now_ - tprev;                                            \
      tprev = now_;                                                     \
    }                                                                   \
  } while (0)

// ===============================================================================================================
// xf_tail
// ===============================================================================================================
template <typename T, int DBG, int CC>  // DBG 0 = production, 1 = intermediate dumps (tests), 2 = section timing (tools/probes/xf_anatomy.py)
__global__ __launch_bounds__(XNT) void xf_tail_kernel(const XfParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  XF_CFG(CC);
  constexpr bool act_ = true;  // (shadowed inside the GEGLU projection runs: waves that own none of a run's blocks)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // C = 320: (wm, wn) = (wave / 2, wave % 2); C = 640: (wave % 2, wave / 2) — the waves of one column-group pair
  // (wn / 2) then sit on four different SIMDs, which a GEGLU projection run relies on
  const int wm = XWN == 2 ? wave >> 1 : wave & 1, wn = XWN == 2 ? wave & 1 : wave >> 1;
  const int lq = lane & 31, hi = lane >> 5;
  const int lane16 = lane * 16;

  // this workgroup's panels: XCD x owns panels [x*q, (x+1)*q), workgroup `loc` of the XCD takes loc, loc + gx, ...
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int x0 = xcd * p.q;
  int xn = p.npanels - x0;
  xn = xn > p.q ? p.q : xn;
  const int nmine = xn > loc ? (xn - loc + p.gx - 1) / p.gx : 0;
  if (nmine == 0) return;
  const int total = nmine * TAIL_TILES;

  const __amdgpu_buffer_rsrc_t w_srd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wstream), 0, TAIL_TILES * TILE_BYTES, 0x00020000);
  const long long src_rows = p.pair_bs ? (long long)p.M / 2 : (long long)p.M;
  const __amdgpu_buffer_rsrc_t o_srd = xsrd(p.o, (((src_rows - 1) * p.ldo + XC) * 2 + 15) & ~15LL);
  const __amdgpu_buffer_rsrc_t h_srd = xsrd(p.h, ((src_rows - 1) * p.ldh + XC) * 2);
  const __amdgpu_buffer_rsrc_t x_srd = xsrd(p.x, ((src_rows - 1) * p.ldx + XC) * 2);
  const __amdgpu_buffer_rsrc_t out_srd = xsrd(p.out, (((long long)p.M - 1) * p.ldout + XC) * 2);
  const __amdgpu_buffer_rsrc_t prm_srd = xsrd(p.prm, 5 * XC * 4);
  const long long nsamp = (long long)p.M / p.L;
  const __amdgpu_buffer_rsrc_t kf_srd = xsrd(p.kf, nsamp * XHEADS * (XKB * 4) * 1024);
  const __amdgpu_buffer_rsrc_t vf_srd = xsrd(p.vf, nsamp * XHEADS * (2 * 6) * 1024);
  const int w_voff = tid * 16;
  const int pan_voff = (int)((lq * p.ldo + hi * 8) * 2);
  const int prm_voff = (160 * wn + 4 * hi) * 4;

  int issued = 0, consumed = 0, landed = 0, s_slot = 0, c_slot = 0, s_t = 0;
  const char* tbase = smem;
  // ablation instantiations (wall-clock A/B, tools/probes/xf_anatomy.py): 3 = no staging, 4 = no MFMAs, 5 = no fragment reads
  constexpr int abl = DBG == 3 ? 8 : (DBG == 4 ? 16 : (DBG == 5 ? 32 : 0));
  XF_STAGE(TAIL_TILES);
  XF_STAGE(TAIL_TILES);

  f32x16 acc[5];
  typename T::vec8 xfr[2] = {}, wfr[2][5] = {};
  uint2 hres[5][4];
  unsigned long long ta[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  if (DBG == 2) tprev = __builtin_amdgcn_s_memtime();

  for (int pi = 0; pi < nmine; ++pi) {
    const int panel = x0 + loc + pi * p.gx;
    const long long row0 = (long long)panel * XBM;
    const int b = (int)(row0 / p.L);   // sample of the (full) batch; a panel never straddles samples (L % 128 == 0)
    long long srow0 = row0;
    if (p.pair_bs) {
      const int g2 = b / (2 * p.pair_bs), r2 = b % p.pair_bs;
      srow0 = ((long long)(g2 * p.pair_bs + r2)) * p.L + (row0 - (long long)b * p.L);
    }
    // ---------------- phase 0: X = attention output panel; acc = h + b_out1 ----------------
    xbarrier();  // previous panel: every wave is done with X and the GEGLU chunk / LayerNorm scratch
    XF_LOAD_PANEL(o_srd, p.ldo, srow0);
    {
      uint4 res[5][2];
      XF_RES_LOAD(res, h_srd, p.ldh, srow0);
      XF_ACC_BIAS(0);
      XF_ACC_ADD_RES(res);
    }
    xwait<0>();
    landed = issued;
    XF_TS(0);
    // ---------------- phase 1: h1 = attn @ Wo1^T + b + h;  X = LayerNorm2(h1) ----------------
    XF_GEMMCC(TAIL_TILES);
    XF_TS(1);
    XF_ROUND_TO_HRES();
    if (DBG == 1 && p.stop_after == 11) { XF_DUMP_HRES(); XF_SKIP_REST(TAIL_TILES); continue; }
    XF_LAYERNORM_TO_X();
    XF_ACC_BIAS(1);  // q = LN2(h1) Wq^T: beta2 Wq^T (the folded LayerNorm shift) is the accumulators' initial value
    xwait<0>();
    landed = issued;
    if (DBG == 1 && p.stop_after == 1) { XF_DUMP_X(); XF_SKIP_REST(TAIL_TILES); continue; }
    XF_TS(2);
    // ---------------- phase 2: q = LN2(h1) @ Wq^T  -> X ----------------
    XF_GEMMCC(TAIL_TILES);
    XF_TS(3);
    uint4 kfr[12];  // context K fragments of this wave's first cross-attention unit (rowblk wave / 5, head wave % 5)
    {
      const int kf_so = (b * XHEADS + wave % XHEADS) * (XKB * 4) * 1024;
#pragma unroll
      for (int i = 0; i < 12; ++i) kfr[i] = xld_u4(kf_srd, lane16, kf_so + i * 1024);
    }
    xbarrier();  // all waves are done reading X
    XF_STORE_X(acc);
    if (DBG == 1 && p.stop_after == 2) { XF_DUMP_X(); XF_SKIP_REST(TAIL_TILES); continue; }
    xbarrier();
    XF_TS(4);
    // ---------------- phase 3: text cross-attention, all heads, in place in X ----------------
    // (K fragments of a unit are fetched one unit ahead — the first ones under the q store above — and V^T fragments
    // behind the S^T MFMAs, under the softmax arithmetic: the loads are L2 hits of ~1 us that stood in front of every
    // MFMA group before: 13 us per panel)
    for (int u = wave; u < XWM * XHEADS; u += 8) {
      const int rb = u / XHEADS, hd = u - rb * XHEADS;
      typename T::vec8 qf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const typename T::vec8*>(smem + (rb * XKS + 4 * hd + ks) * 1024 + lane16);
      const int vf_so = (b * XHEADS + hd) * (2 * 6) * 1024;
      int hi4 = 4 * hi, ob = xoff(rb, XKS, 64 * hd + 4 * hi, lq);  // opaque: keeps the 48 key compares / 8 store offsets
      XF_OPAQUE(hi4);                                              // of a unit from being hoisted out of the panel loop
      XF_OPAQUE(ob);
      f32x16 s_acc[XKB];
#pragma unroll
      for (int kb = 0; kb < XKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          s_acc[kb] = T::mfma32(__builtin_bit_cast(typename T::vec8, kfr[kb * 4 + ks]), qf[ks], s_acc[kb]);
      }
      uint4 vfr[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) vfr[i] = xld_u4(vf_srd, lane16, vf_so + i * 1024);
      if (u + 8 < XWM * XHEADS) {  // next unit of this wave
        const int kf_so = (b * XHEADS + (u + 8) % XHEADS) * (XKB * 4) * 1024;
#pragma unroll
        for (int i = 0; i < 12; ++i) kfr[i] = xld_u4(kf_srd, lane16, kf_so + i * 1024);
      }
      float mx = -1e30f;
#pragma unroll
      for (int kb = 0; kb < XKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + hi4;
          const float sv = key < p.Lk ? s_acc[kb][r] : -1e30f;
          s_acc[kb][r] = sv;
          mx = fmaxf(mx, sv);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float neg_m = -mx * p.c;
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < XKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][r], p.c, neg_m));
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
        for (int t = 0; t < 2; ++t)
          o_acc[t] = T::mfma32(__builtin_bit_cast(typename T::vec8, vfr[t * 6 + s]), pfrag, o_acc[t]);
      }
      const float inv = 1.0f / psum;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 pk;
          pk.x = T::pack2(o_acc[t][4 * g + 0] * inv, o_acc[t][4 * g + 1] * inv);
          pk.y = T::pack2(o_acc[t][4 * g + 2] * inv, o_acc[t][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(smem + ob + (2 * t + (g >> 1)) * 1024 + (g & 1) * 512) = pk;
        }
    }
    XF_TS(5);
    // acc = h1 + b_out2 (the K / V fragment loads above are ordinary loads: drain, then the ring is known landed)
    XF_ACC_BIAS(2);
    XF_ACC_ADD_HRES();
    xwait<0>();
    landed = issued;
    if (DBG == 1 && p.stop_after == 3) { XF_DUMP_X(); XF_SKIP_REST(TAIL_TILES); continue; }
    XF_TS(6);
    // ---------------- phase 4: h2 = a @ Wo2^T + b + h1;  X = LayerNorm3(h2) ----------------
    XF_GEMMCC(TAIL_TILES);
    XF_TS(7);
    XF_ROUND_TO_HRES();
    if (DBG == 1 && p.stop_after == 14) { XF_DUMP_HRES(); XF_SKIP_REST(TAIL_TILES); continue; }
    XF_LAYERNORM_TO_X();
    // acc = h2 + b_ff2
    XF_ACC_BIAS(3);
    XF_ACC_ADD_HRES();
    xwait<0>();
    landed = issued;
    if (DBG == 1 && p.stop_after == 4) { XF_DUMP_X(); XF_SKIP_REST(TAIL_TILES); continue; }
    XF_TS(8);
    // ---------------- phase 5: GEGLU feed-forward, 20 chunks of 32 WN hidden units ----------------
    for (int cch = 0; cch < XCH; ++cch) {
      f32x16 gacc[2];
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        // run `sub` holds the (value, gate) blocks of the column groups 2 sub, 2 sub + 1: the other waves only keep the ring
        // moving (acquire / refill) — the run is bound by the weight stream either way
        const bool act_ = NSUB == 1 || (wn >> 1) == sub;
        const int wl = wn & 1;
        XF_RUN_ACQ(2, XF_A_X5, 2 * wl);
        if (act_) {  // G := GEGLU projection bias (f32 side data of the run's first tile): [wl][value | gate][32]
          const float* bz = reinterpret_cast<const float*>(tbase + TILE_W) + wl * 64 + 4 * hi;
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 b_ = *reinterpret_cast<const float4*>(bz + blk * 32 + 8 * g);
              gacc[blk][4 * g + 0] = b_.x;
              gacc[blk][4 * g + 1] = b_.y;
              gacc[blk][4 * g + 2] = b_.z;
              gacc[blk][4 * g + 3] = b_.w;
            }
        }
        if (issued < total) XF_STAGE(TAIL_TILES);  // (behind the side-data reads: hipcc drains vmcnt in front of an LDS read
                                                   // that follows a direct-to-LDS load it cannot tell apart from the ring)
        XF_RUN_BODY(F1T, 5, 2, 4, XF_A_X5, 2 * wl, gacc, TAIL_TILES);
      }
      XF_TS(9);
      // g = value * gelu(gate) -> GEGLU chunk image [rowblk WM][kstep GKST] (its previous readers passed a barrier since)
      int gb = GB_OFF + xoff(wm, GKST, 32 * wn + 4 * hi, lq);
      XF_OPAQUE(gb);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gacc[0][4 * g + e] * gelu_fast(gacc[1][4 * g + e]);
        uint2 pk;
        pk.x = T::pack2(v[0], v[1]);
        pk.y = T::pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(smem + gb + (g >> 1) * 1024 + (g & 1) * 512) = pk;
      }
      XF_TS(10);
      XF_RUN_BEGIN(5, XF_A_GB, 5 * wn, TAIL_TILES);  // its barrier publishes the chunk
      XF_RUN_BODY(F2T, GNKS, 5, XNB, XF_A_GB, 5 * wn, acc, TAIL_TILES);
      XF_TS(11);
    }
    // ---------------- phase 6: h3 -> X; acc = x + b_po; out = h3 @ Wpo^T + ... ----------------
    xbarrier();  // all waves are done reading X (LN3 output) — the last FF1 tile was many barriers ago, kept for clarity
    XF_STORE_X(acc);
    if (DBG == 1 && p.stop_after == 5) { XF_DUMP_X(); XF_SKIP_REST(TAIL_TILES); continue; }
    {
      uint4 res[5][2];
      XF_RES_LOAD(res, x_srd, p.ldx, srow0);
      XF_ACC_BIAS(4);
      XF_ACC_ADD_RES(res);
    }
    xwait<0>();
    landed = issued;
    XF_TS(12);
    XF_GEMMCC(TAIL_TILES);
    XF_TS(13);
    XF_ROW_STORE(out_srd, p.ldout, row0, 0);
    XF_TS(14);
  }
  if (DBG == 2 && wave == 0 && lane == 0) {  // (the timing build's section table goes to the tensor passed as `h`)
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<u16*>(p.h)) + (long long)blockIdx.x * 16;
    for (int i = 0; i < 16; ++i) dbg[i] = ta[i];
  }
#endif
}

// ===============================================================================================================
// xf_head: x -> GroupNorm apply -> proj_in -> h (stored) -> LayerNorm1 -> q | k (stored [M, 2C]) and v^T (stored
// transposed per sample for the flash-attention kernel)
// ===============================================================================================================
template <typename T, int CC>
__global__ __launch_bounds__(XNT) void xf_head_kernel(const XfParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  XF_CFG(CC);
  constexpr bool act_ = true;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = XWN == 2 ? wave >> 1 : wave & 1, wn = XWN == 2 ? wave & 1 : wave >> 1;
  const int lq = lane & 31, hi = lane >> 5;
  const int lane16 = lane * 16;

  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int x0 = xcd * p.q;
  int xn = p.npanels - x0;
  xn = xn > p.q ? p.q : xn;
  const int nmine = xn > loc ? (xn - loc + p.gx - 1) / p.gx : 0;
  if (nmine == 0) return;
  const int total = nmine * HEAD_TILES;

  const __amdgpu_buffer_rsrc_t w_srd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wstream), 0, HEAD_TILES * TILE_BYTES, 0x00020000);
  const int w_voff = tid * 16;

  int issued = 0, consumed = 0, landed = 0, s_slot = 0, c_slot = 0, s_t = 0;
  const char* tbase = smem;
  constexpr int abl = 0;
  XF_STAGE(HEAD_TILES);
  XF_STAGE(HEAD_TILES);

  const __amdgpu_buffer_rsrc_t xin_srd = xsrd(p.o, (((long long)p.M - 1) * p.ldo + XC) * 2);
  const __amdgpu_buffer_rsrc_t h_srd = xsrd(p.h, (((long long)p.M - 1) * p.ldh + XC) * 2);
  const __amdgpu_buffer_rsrc_t out_srd = xsrd(p.out, (((long long)p.M - 1) * p.ldout + 2 * XC) * 2);
  const __amdgpu_buffer_rsrc_t prm_srd = xsrd(p.prm, 4 * XC * 4);
  const long long nsamp = (long long)p.M / p.L;
  const __amdgpu_buffer_rsrc_t ab_srd = xsrd(p.ab, nsamp * 2 * XC * 4);
  const __amdgpu_buffer_rsrc_t vt_srd = xsrd(p.vt, ((nsamp - 1) * p.vt_bs + (long long)(XC - 1) * p.vt_ld + p.L) * 2);
  const int prm_voff = (160 * wn + 4 * hi) * 4;
  const int xin_voff = (int)((lq * p.ldo + hi * 8) * 2);

  f32x16 acc[5];
  typename T::vec8 xfr[2], wfr[2][5];

  for (int pi = 0; pi < nmine; ++pi) {
    const int panel = x0 + loc + pi * p.gx;
    const long long row0 = (long long)panel * XBM;
    const int b = (int)(row0 / p.L);
    const int l0 = (int)(row0 - (long long)b * p.L);
    // ---------------- phase 0: X = GroupNorm(x) = x * a[b, c] + s[b, c] (statistics precomputed) ----------------
    xbarrier();  // previous panel: every wave is done with X
    {
      // pass i of wave w fills fragment piece 8 i + w (rowblk, kstep); a lane owns 8 channels of one row
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int piece = 8 * i + wave;
        const int rb = piece / XKS, ks = piece - rb * XKS;
        const uint4 v = xld_u4(xin_srd, xin_voff, (int)(((row0 + rb * 32) * p.ldo + ks * 16) * 2));
        const int ab_so = (b * 2 * XC + ks * 16) * 4;
        const float4 a0 = xld_f4(ab_srd, hi * 32, ab_so), a1 = xld_f4(ab_srd, hi * 32 + 16, ab_so);
        const float4 s0 = xld_f4(ab_srd, hi * 32, ab_so + XC * 4), s1 = xld_f4(ab_srd, hi * 32 + 16, ab_so + XC * 4);
        float f[8];
        unpack8<T>(v, f);
        f[0] = f[0] * a0.x + s0.x; f[1] = f[1] * a0.y + s0.y; f[2] = f[2] * a0.z + s0.z; f[3] = f[3] * a0.w + s0.w;
        f[4] = f[4] * a1.x + s1.x; f[5] = f[5] * a1.y + s1.y; f[6] = f[6] * a1.z + s1.z; f[7] = f[7] * a1.w + s1.w;
        *reinterpret_cast<uint4*>(smem + piece * 1024 + lane16) = pack8<T>(f);
      }
    }
    XF_ACC_BIAS(0);
    xwait<0>();
    landed = issued;
    // ---------------- phase 1: h = GN(x) @ Wpi^T + b -> HBM;  X = LayerNorm1(h) ----------------
    XF_GEMMCC(HEAD_TILES);
    XF_ROW_STORE(h_srd, p.ldh, row0, 0);
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = T::to_f32(T::from_f32(acc[j][r]));
    XF_LAYERNORM_TO_X();
    xwait<0>();
    landed = issued;
    // ---------------- phase 2 / 3: q, k (bias = the folded LayerNorm shift) -> out[:, 0:C], out[:, C:2C] ----------------
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      XF_ACC_BIAS(1 + part);
      XF_GEMMCC(HEAD_TILES);
      XF_ROW_STORE(out_srd, p.ldout, row0, part * XC);
      xwait<0>();
      landed = issued;
    }
    // ---------------- phase 4: v -> transposed through LDS -> v^T[b, c, l0 .. l0 + BM) ----------------
    XF_ACC_BIAS(3);
    XF_GEMMCC(HEAD_TILES);
    xbarrier();  // all waves are done reading X: reuse it as the [C channels][BM rows] transpose buffer
    {
      u16* ts = reinterpret_cast<u16*>(smem);
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = 160 * wn + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hi;
          ts[n * XBM + 32 * wm + lq] = T::from_f32(acc[j][r]);
        }
      xbarrier();
      const int vt_so = (int)(((long long)b * p.vt_bs + l0) * 2);
      for (int q = tid; q < XC * (XBM / 8); q += XNT) {
        const int n = q / (XBM / 8), mc = q % (XBM / 8);
        const uint4 v = *reinterpret_cast<const uint4*>(ts + n * XBM + mc * 8);
        xst_u4(vt_srd, (int)((n * p.vt_ld + mc * 8) * 2) + vt_so, v);
      }
    }
    xwait<0>();
    landed = issued;
  }
#endif
}

int g_xf_variant = 1;

int xf_grid(int npanels, int* q, int* gx) {
  *q = cdiv(npanels, 8);
  *gx = *q < 32 ? *q : 32;  // one workgroup per CU (the panel + ring take the whole LDS)
  return 8 * *gx;
}

template <typename KT>
int xf_set_lds(KT kern) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, XF_LDS);
}

template <int CC>
int xf_launch_tail(int dtype, int stop_after, int grid, hipStream_t s, const XfParams& p) {
  static bool attr_set = false;
  if (!attr_set) {
    if (xf_set_lds(&xf_tail_kernel<F16, 0, CC>) != 0 || xf_set_lds(&xf_tail_kernel<BF16, 0, CC>) != 0 ||
        xf_set_lds(&xf_tail_kernel<F16, 1, CC>) != 0 || xf_set_lds(&xf_tail_kernel<BF16, 1, CC>) != 0
#ifdef DBIR_DIAG
        || xf_set_lds(&xf_tail_kernel<F16, 2, CC>) != 0 || xf_set_lds(&xf_tail_kernel<F16, 3, CC>) != 0 ||
        xf_set_lds(&xf_tail_kernel<F16, 4, CC>) != 0 || xf_set_lds(&xf_tail_kernel<F16, 5, CC>) != 0
#endif
    ) {
      dbir_set_error("dbir_xf_tail: cannot reserve %d bytes of LDS", XF_LDS);
      return DBIR_ERR_LAUNCH;
    }
    attr_set = true;
  }
  if (stop_after >= 99) {  // section timing (99) / ablations (103 - 105), f16 only, results meaningless
#ifdef DBIR_DIAG           // only in a `DBIR_DIAG=1 sh build.sh` library (tools/probes/xf_anatomy.py), never in the production ABI
    DBIR_CHECK_ARG(dtype == DBIR_F16, "dbir_xf_tail: the timing instantiations are f16 only");
    if (stop_after == 99) hipLaunchKernelGGL((xf_tail_kernel<F16, 2, CC>), dim3(grid), dim3(XNT), XF_LDS, s, p);
    else if (stop_after == 103) hipLaunchKernelGGL((xf_tail_kernel<F16, 3, CC>), dim3(grid), dim3(XNT), XF_LDS, s, p);
    else if (stop_after == 104) hipLaunchKernelGGL((xf_tail_kernel<F16, 4, CC>), dim3(grid), dim3(XNT), XF_LDS, s, p);
    else hipLaunchKernelGGL((xf_tail_kernel<F16, 5, CC>), dim3(grid), dim3(XNT), XF_LDS, s, p);
#else
    dbir_set_error("dbir_xf_tail: stop_after %d (timing / ablation instantiations) needs a DBIR_DIAG build", stop_after);
    return DBIR_ERR_ARG;
#endif
  } else if (stop_after) {  // debug instantiation (tests): intermediate dumps
    if (dtype == DBIR_F16) hipLaunchKernelGGL((xf_tail_kernel<F16, 1, CC>), dim3(grid), dim3(XNT), XF_LDS, s, p);
    else hipLaunchKernelGGL((xf_tail_kernel<BF16, 1, CC>), dim3(grid), dim3(XNT), XF_LDS, s, p);
  } else {
    if (dtype == DBIR_F16) hipLaunchKernelGGL((xf_tail_kernel<F16, 0, CC>), dim3(grid), dim3(XNT), XF_LDS, s, p);
    else hipLaunchKernelGGL((xf_tail_kernel<BF16, 0, CC>), dim3(grid), dim3(XNT), XF_LDS, s, p);
  }
  return DBIR_OK;
}

template <int CC>
int xf_launch_head(int dtype, int grid, hipStream_t s, const XfParams& p) {
  static bool attr_set = false;
  if (!attr_set) {
    if (xf_set_lds(&xf_head_kernel<F16, CC>) != 0 || xf_set_lds(&xf_head_kernel<BF16, CC>) != 0) {
      dbir_set_error("dbir_xf_head: cannot reserve %d bytes of LDS", XF_LDS);
      return DBIR_ERR_LAUNCH;
    }
    attr_set = true;
  }
  if (dtype == DBIR_F16) hipLaunchKernelGGL((xf_head_kernel<F16, CC>), dim3(grid), dim3(XNT), XF_LDS, s, p);
  else hipLaunchKernelGGL((xf_head_kernel<BF16, CC>), dim3(grid), dim3(XNT), XF_LDS, s, p);
  return DBIR_OK;
}

}  // namespace

void dbir_xf_set_variant(int v) { g_xf_variant = v; }

// second-generation kernels (csrc/xformer2.hip): selected by the LENGTH of the packed weight stream (diffbir_amd/xformer.py packs one
// format or the other; the two lengths differ for every C)
extern "C" int dbir_xf2_geometry(int C, int* panel_rows, long long* head_bytes, long long* tail_bytes, int* tail_prm_floats);
int dbir_xf2_tail_impl(int dtype, const void* attn_out, long long ldo, const void* h, long long ldh, const void* x, long long ldx,
                       void* out, long long ldout, int M, int L, int C, int pair_bs, const void* wstream, const float* prm,
                       const void* kfrag, const void* vfrag, int Lk, float scale, int stop_after, void* stream);
int dbir_xf2_head_impl(int dtype, const void* x, long long ldx, const float* gn_scale_shift, void* h, long long ldh, void* qk,
                       long long ldqk, void* vt, long long vt_ld, long long vt_bstride, int M, int L, int C, const void* wstream,
                       const float* prm, void* stream);

extern "C" int dbir_xf_tile_bytes(void) { return TILE_BYTES; }
extern "C" int dbir_xf_tail_tiles(void) { return XfCfg<320>::TAIL_TILES; }
extern "C" int dbir_xf_head_tiles(void) { return XfCfg<320>::HEAD_TILES; }
// geometry of the fused kernels for inner width C (320 or 640): panel rows, tiles of the head / tail weight streams
extern "C" int dbir_xf_geometry(int C, int* panel_rows, int* head_tiles, int* tail_tiles) {
  DBIR_CHECK_ARG(C == 320 || C == 640, "dbir_xf_geometry: the fused transformer kernels are built for C = 320 and 640 (got %d)", C);
  if (panel_rows) *panel_rows = C == 320 ? XfCfg<320>::BM : XfCfg<640>::BM;
  if (head_tiles) *head_tiles = C == 320 ? XfCfg<320>::HEAD_TILES : XfCfg<640>::HEAD_TILES;
  if (tail_tiles) *tail_tiles = C == 320 ? XfCfg<320>::TAIL_TILES : XfCfg<640>::TAIL_TILES;
  return DBIR_OK;
}

extern "C" int dbir_xf_tail(int dtype, const void* attn_out, long long ldo, const void* h, long long ldh, const void* x,
                            long long ldx, void* out, long long ldout, int M, int L, int C, int pair_bs,
                            const void* wstream, long long wstream_bytes, const float* prm, const void* kfrag,
                            const void* vfrag, int Lk, float scale, int stop_after, void* stream) {
  DBIR_CHECK_ARG(attn_out && h && x && out && wstream && prm && kfrag && vfrag, "dbir_xf_tail: null pointer");
  DBIR_CHECK_ARG(dtype == DBIR_F16 || dtype == DBIR_BF16, "dbir_xf_tail: bad dtype %d", dtype);
  DBIR_CHECK_ARG(C == 320 || C == 640, "dbir_xf_tail: built for C = 320 and 640 (got %d)", C);
  long long v2_tail_bytes = 0;
  dbir_xf2_geometry(C, nullptr, nullptr, &v2_tail_bytes, nullptr);
  const bool v2 = wstream_bytes == v2_tail_bytes;
  const int BM = C == 320 ? XfCfg<320>::BM : XfCfg<640>::BM;
  const int tiles = C == 320 ? XfCfg<320>::TAIL_TILES : XfCfg<640>::TAIL_TILES;
  DBIR_CHECK_ARG(M > 0 && L > 0 && L % BM == 0 && M % L == 0, "dbir_xf_tail: M %d must be whole samples of L %d rows, L %% %d == 0", M, L, BM);
  DBIR_CHECK_ARG(pair_bs >= 0 && (pair_bs == 0 || (M / L) % (2 * pair_bs) == 0), "dbir_xf_tail: bad pair_bs %d for %d samples", pair_bs, M / L);
  DBIR_CHECK_ARG(ldo % 8 == 0 && ldh % 8 == 0 && ldx % 8 == 0 && ldout % 8 == 0 && ldo >= C && ldh >= C && ldx >= C && ldout >= C,
                 "dbir_xf_tail: row strides must be multiples of 8 and >= C");
  DBIR_CHECK_ARG(((reinterpret_cast<uintptr_t>(attn_out) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(x) |
                   reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(wstream) | reinterpret_cast<uintptr_t>(prm) |
                   reinterpret_cast<uintptr_t>(kfrag) | reinterpret_cast<uintptr_t>(vfrag)) & 15) == 0,
                 "dbir_xf_tail: pointers must be 16-byte aligned");
  DBIR_CHECK_ARG(v2 || wstream_bytes >= (long long)tiles * TILE_BYTES, "dbir_xf_tail: weight stream too short");
  DBIR_CHECK_ARG(Lk > 0 && Lk <= XLKP, "dbir_xf_tail: context length %d > %d", Lk, XLKP);
  const long long src_rows = pair_bs ? M / 2 : M;
  DBIR_CHECK_ARG(((src_rows - 1) * ldo + C) * 2 < 0x7ffffe00LL && ((src_rows - 1) * ldh + C) * 2 < 0x7ffffe00LL &&
                     ((src_rows - 1) * ldx + C) * 2 < 0x7ffffe00LL && ((long long)(M - 1) * ldout + C) * 2 < 0x7ffffe00LL,
                 "dbir_xf_tail: a tensor is too large for a 2 GB buffer descriptor");
  if (v2)
    return dbir_xf2_tail_impl(dtype, attn_out, ldo, h, ldh, x, ldx, out, ldout, M, L, C, pair_bs, wstream, prm, kfrag, vfrag, Lk, scale,
                              stop_after, stream);
  XfParams p;
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
  p.variant = g_xf_variant;
  p.prm_row_bytes = C * 4;
  const int grid = xf_grid(p.npanels, &p.q, &p.gx);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rc = C == 320 ? xf_launch_tail<320>(dtype, stop_after, grid, s, p) : xf_launch_tail<640>(dtype, stop_after, grid, s, p);
  if (rc != DBIR_OK) return rc;
  DBIR_CHECK_LAUNCH("dbir_xf_tail");
  return DBIR_OK;
}

extern "C" int dbir_xf_head(int dtype, const void* x, long long ldx, const float* gn_scale_shift, void* h, long long ldh,
                            void* qk, long long ldqk, void* vt, long long vt_ld, long long vt_bstride, int M, int L, int C,
                            const void* wstream, long long wstream_bytes, const float* prm, void* stream) {
  DBIR_CHECK_ARG(x && gn_scale_shift && h && qk && vt && wstream && prm, "dbir_xf_head: null pointer");
  DBIR_CHECK_ARG(dtype == DBIR_F16 || dtype == DBIR_BF16, "dbir_xf_head: bad dtype %d", dtype);
  DBIR_CHECK_ARG(C == 320 || C == 640, "dbir_xf_head: built for C = 320 and 640 (got %d)", C);
  long long v2_head_bytes = 0;
  dbir_xf2_geometry(C, nullptr, &v2_head_bytes, nullptr, nullptr);
  const bool v2 = wstream_bytes == v2_head_bytes;
  const int BM = C == 320 ? XfCfg<320>::BM : XfCfg<640>::BM;
  const int tiles = C == 320 ? XfCfg<320>::HEAD_TILES : XfCfg<640>::HEAD_TILES;
  DBIR_CHECK_ARG(M > 0 && L > 0 && L % BM == 0 && M % L == 0, "dbir_xf_head: M %d must be whole samples of L %d rows, L %% %d == 0", M, L, BM);
  DBIR_CHECK_ARG(ldx % 8 == 0 && ldh % 8 == 0 && ldqk % 8 == 0 && vt_ld % 8 == 0 && vt_bstride % 8 == 0 && ldx >= C && ldh >= C &&
                     ldqk >= 2 * C && vt_ld >= L,
                 "dbir_xf_head: bad strides");
  DBIR_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(qk) |
                   reinterpret_cast<uintptr_t>(vt) | reinterpret_cast<uintptr_t>(wstream) | reinterpret_cast<uintptr_t>(prm) |
                   reinterpret_cast<uintptr_t>(gn_scale_shift)) & 15) == 0,
                 "dbir_xf_head: pointers must be 16-byte aligned");
  DBIR_CHECK_ARG(v2 || wstream_bytes >= (long long)tiles * TILE_BYTES, "dbir_xf_head: weight stream too short");
  DBIR_CHECK_ARG(((long long)(M - 1) * ldx + C) * 2 < 0x7ffffe00LL && ((long long)(M - 1) * ldh + C) * 2 < 0x7ffffe00LL &&
                     ((long long)(M - 1) * ldqk + 2 * C) * 2 < 0x7ffffe00LL &&
                     ((long long)(M / L - 1) * vt_bstride + (long long)(C - 1) * vt_ld + L) * 2 < 0x7ffffe00LL,
                 "dbir_xf_head: a tensor is too large for a 2 GB buffer descriptor");
  if (v2) return dbir_xf2_head_impl(dtype, x, ldx, gn_scale_shift, h, ldh, qk, ldqk, vt, vt_ld, vt_bstride, M, L, C, wstream, prm, stream);
  XfParams p;
  memset(&p, 0, sizeof(p));
  p.o = (const u16*)x; p.ldo = ldx;
  p.h = (const u16*)h; p.ldh = ldh;
  p.out = (u16*)qk; p.ldout = ldqk;
  p.vt = (u16*)vt; p.vt_ld = vt_ld; p.vt_bs = vt_bstride;
  p.ab = gn_scale_shift;
  p.M = M; p.L = L;
  p.wstream = wstream; p.prm = prm;
  p.npanels = M / BM;
  p.variant = g_xf_variant;
  p.prm_row_bytes = C * 4;
  const int grid = xf_grid(p.npanels, &p.q, &p.gx);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rc = C == 320 ? xf_launch_head<320>(dtype, grid, s, p) : xf_launch_head<640>(dtype, grid, s, p);
  if (rc != DBIR_OK) return rc;
  DBIR_CHECK_LAUNCH("dbir_xf_head");
  return DBIR_OK;
}
