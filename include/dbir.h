/* dbir.h — C ABI of the MI355X (gfx950) DiffBIR hot-path kernels (libdbir_hip.so).
 *
 * The reference (XPixelGroup/DiffBIR) is pure Python on PyTorch: it has no FFI of its own.  Its lowest
 * boundary on the hot path is the set of torch.nn.functional calls listed below;
 * each entry point here replaces one of those call families and cites the reference call sites it
 * stands in for.  All pointers are raw DEVICE pointers owned by the caller (PyTorch allocations), sizes
 * are plain ints, `stream` is a hipStream_t passed as void*.  Every function only enqueues work on
 * `stream` (no hidden synchronisation, no allocation) and returns 0 on success or a DBIR_ERR_* code;
 * `dbir_last_error()` returns a thread-local message.  No torch types cross this boundary.
 *
 * Activation layout inside the engine: channels-last (NHWC), 16-bit (f16 or bf16, `dtype`), so that a
 * feature map is directly the row-major [B*H*W, C] A-operand of an MFMA GEMM.
 */
#ifndef DBIR_H
#define DBIR_H
#ifdef __cplusplus
extern "C" {
#endif

#define DBIR_OK 0
#define DBIR_ERR_ARG 1
#define DBIR_ERR_LAUNCH 2

#define DBIR_F16 0
#define DBIR_BF16 1

/* epilogue activations */
#define DBIR_ACT_NONE 0
#define DBIR_ACT_SILU 1
#define DBIR_ACT_GELU 2   /* exact erf GELU (F.gelu default) */
#define DBIR_ACT_LRELU 3  /* LeakyReLU(act_param) */
#define DBIR_ACT_GEGLU 4  /* x * gelu(gate): weights packed value/gate interleaved per 32 columns */

#define DBIR_MODE_LINEAR 0
#define DBIR_MODE_CONV3X3 1

const char* dbir_last_error(void);
/* 3 since round 3: dbir_gemm takes a non-const descriptor (stats / stats_rows fields at its end), dbir_xf_head / dbir_xf_tail /
 * dbir_xf_geometry, dbir_groupnorm_affine, dbir_groupnorm_from_partials; tiles 80 - 89 retired, 90 - 92 added.
 * 4 since round 4: dbir_gemm_desc.stats holds [sum, M2] per (row tile, column) instead of [sum, sum of squares] (and
 * dbir_groupnorm_from_partials reads that), split-K launches emit them; tile 80 = the fine-phase 256x320 kernel.
 * 5 since round 5: dbir_gemm_desc.upsample == 2 (parity-collapsed upsample convolution), dbir_groupnorm_apply_partials,
 * dbir_copy_rows, dbir_plan_* / dbir_cldm_forward (module-level entry point: a recorded network evaluation replayed from C). */
int dbir_abi_version(void);
/* Process-wide tuning / A-B switches (never needed for correctness): DBIR_OPT_ATTN_VARIANT 2 (default) = LDS-resident
 * cross-attention kernel for Lk <= 96 + generic flash kernel otherwise, 3 = generic flash kernel for every shape, 4 / 5 =
 * the generic kernel's pre-round-4 softmax compiled for 4 / 3 resident waves per SIMD, 6 = pre-round-4 softmax (scale and
 * running maximum applied per score by a v_fma, f32 row sum) at the default register budget, 8 = long self-attentions on the 8-wave
 * attn3_kernel whose wave groups alternate matrix / softmax phases (round-6 experiment, measured 0.65x: profiles/r6_attn3_ab.txt);
 * 1000 + n = attn3's query-row threshold. */
#define DBIR_OPT_ATTN_VARIANT 1
/* DBIR_OPT_XF_VARIANT: weight staging schedule of dbir_xf_head / dbir_xf_tail — 1 (default) = the two wave groups stage
 * their shares of a tile at opposite ends of the tile's MFMAs, 0 = every wave stages first (A/B).  First-generation kernels only
 * (csrc/xformer.hip); the second-generation kernels (csrc/xformer2.hip) have no staging to schedule. */
#define DBIR_OPT_XF_VARIANT 2
int dbir_set_option(int key, int value);

/* ---------------------------------------------------------------------------------------------
 * dbir_gemm — fused implicit-GEMM on MFMA: C = epilogue(A (*) W^T)
 * Replaces: F.linear (reference attention.py:22,41,67-73,310,331; unet.py:168,477-479; swinir.py:23-25,
 * 111,113), F.conv2d 1x1 (unet.py:189; controlnet.py:311; vae.py:93,128-139,569-570) and F.conv2d 3x3
 * stride 1/2 incl. the preceding nearest-x2 F.interpolate (unet.py:76,99-101,152,178; vae.py:34,46-54,77,84;
 * swinir.py:471,768,798-808,879-884), plus the elementwise ops the reference performs around them
 * (bias, `h + emb_out` unet.py:221, residual adds, GEGLU attention.py:24-26, SiLU/GELU/LeakyReLU,
 * `c * scale` cldm.py:164).
 *
 *   mode LINEAR : A is [M, K] 16-bit row-major (row stride lda elements, K % 8 == 0, lda % 8 == 0)
 *   mode CONV3X3: A is an NHWC 16-bit tensor [B, Hi, Wi, Cin] (Cin % 8 == 0); output pixel (b,oy,ox) is row
 *                 m = (b*Ho + oy)*Wo + ox; reduction index k = (ky*3+kx)*Cin + c; input coordinate
 *                 iy = oy*stride - pad + ky (same for x) in the (optionally nearest-x2 upsampled) input.
 *                 upsample == 2 (tile 80 only, round 5): the same nearest-x2 upsample + 3x3 convolution (stride 1, pad 1,
 *                 Hi / Wi powers of two) evaluated as FOUR 2x2 convolutions on the low-resolution input, one per output
 *                 parity (a, b): out[b, 2i+a, 2j+b] = sum_{ty,tx,c} in[b, i+a-1+ty, j+b-1+tx, c] * W[2a+b][n][(ty*2+tx)*Cin+c],
 *                 where W[2a+b] holds the 3x3 taps that land on each low-resolution pixel summed in f32 (4 / 9 of the
 *                 multiplies, exact algebra): K = 4 * Cin, W = [4][Wrows][Kpad]; no residual / row vector.
 *   W           : 16-bit [Wrows, Kpad] row-major (row n = output column n), Kpad % 64 == 0, zero padded.
 *   epilogue    : v = acc + bias[n] + rowvec[(m / rows_per_batch) * rowvec_ld + n]; v = act(v);
 *                 v = v * out_scale + R[m*ldr + n]; store.
 *   store_mode 0: C[m*ldc + n] (16-bit, or f32 if out_f32)
 *   store_mode 1: transposed per batch: C[(m / trans_L) * trans_bstride + n * trans_ld + (m % trans_L)]
 *                 (used to emit V^T for the attention kernel).
 *   gridDim.z = batch with per-z element strides (strideA_z, strideW_z, strideC_z, strideR_z).
 */
typedef struct dbir_gemm_desc {
  int mode, dtype;
  int M, N, K;
  const void* A;
  long long lda, strideA_z;
  const void* W;
  int Wrows, Kpad;
  long long strideW_z;
  /* conv geometry */
  int B, Hi, Wi, Cin, Ho, Wo, stride, pad, upsample;
  /* epilogue */
  const float* bias;
  const void* rowvec; /* 16-bit [*, rowvec_ld] or NULL */
  int rowvec_ld, rows_per_batch;
  int act;
  float act_param, out_scale;
  const void* R; /* 16-bit residual or NULL */
  long long ldr, strideR_z;
  void* C;
  long long ldc, strideC_z;
  int out_f32;
  int store_mode, trans_L;
  long long trans_ld, trans_bstride;
  int batch;
  int tile; /* 0 = auto; generic kernel 1: 128x128, 2: 64x128, 3: 64x64, 4: 128x64; direct-to-LDS kernel 5: 128x128,
               6: 256x64, 7: 64x256, 8/9: 256x128 (4 / 8 waves, 3-stage ring), 10: 256x256, 11: 128x128 4-stage,
               12: 256x128 2-stage; (13: removed in round 2); 14: 256x160 (8 waves), 15: 128x160,
               16: 256x160 (4 waves) — 160-wide tiles fit the UNet's N = 320 k channel counts without padding;
               20 + t for t in {5, 6, 10, 12, 14, 15}: tile t with software-pipelined LDS fragment reads;
               36 / 37 / 38: de-phased two-group 256x128 / 256x160 / 256x64 (3-slot ring, staging of one wave group
               overlaps the MFMAs of the other); 40 / 41: 256x256 with K depth 32 (3-slot de-phased / 4-slot lockstep);
               50 / 51: halo-patch 3x3 convolution kernel (gemm_halo.hip), 256x160 / 256x128 — stride-1 pad-1 convs whose
               256-row tiles are whole image rows (Wo a power of two <= 64): one LDS-resident activation patch per
               64-channel slice serves all 9 taps; split-K slices the channel slices; 52 / 53: the same tiles in lockstep with
               cross-tile fragment prefetch; 54: 256x32 for f32 heads;
               70 - 73: persistent linear kernel (gemm_pers.hip) for the small-K projections — workgroups walk several
               output tiles, the K tiles of all of them form one flat stream through the LDS ring, barrier-free
               register epilogue: 256x160 / 128x160 (two workgroups per CU) / 256x128 / 128x128 (two per CU); dense
               linear with K % 32 == 0, M a multiple of the tile height, N % 8 == 0, no row vector / split-K /
               transposed or f32 store;
               90 - 92: producer / consumer split — four extra loader waves issue every direct-to-LDS copy, the matrix
               waves only read fragments and issue MFMAs (3-slot ring, one barrier per K tile, no split-K): 128x160 (4 matrix
               waves), 128x128 (4), 256x128 (8).
               80: fine-phase 256x320 kernel (gemm_8p.hip): 8 waves of 64x160, K stages of 32 through a 4-slot ring, two
               wave groups one barrier apart (one multiplies while the other reads fragments / issues copies), counted
               vmcnt; linear / 3x3 conv (stride 1 / 2, upsample) with K / Cin % 32 == 0, 16-bit row-major store, no
               GEGLU; split-K is reduced INSIDE the launch (write-through f32 slabs + arrival ticket, last arriver sums in
               slice order) and keeps the statistics stage; ws >= tiles * splitk * 327680 + tiles * 4 bytes.
               93 - 97: register-streaming linear kernel (gemm_rs.hip, round 6): both operands straight into registers with
               16-byte buffer loads (a lane reads 32 contiguous bytes of one row per k-step pair: whole 128-byte lines of the
               row-major operands), 8 waves of 64 x 80 on 16x16x32 MFMAs tiling (RG x 64) x (CG x 80) x KSP k-slices summed
               through LDS in slice order: 93 = 128x160 (2 slices), 94 = 64x160 (4), 95 = 64x80 (8), 96 = 128x320, 97 = 256x160;
               dense linear, K % 64 == 0, M / N multiples of the tile, bias / residual epilogue only.  Measured (profiles/
               r6_rs_ab.txt): 0.55 - 0.65x of the LDS kernels at M = 4096, 1.3x at M = 1024 x K = 1280 (tile 95) — offered to the
               tuner for that class only.
               Ids 60 - 67 exist only in a DBIR_DIAG build (diagnostic ablations, meaningless outputs); 13, 74 - 79 and
               81 - 89 are invalid. */
  /* split-K (direct-to-LDS tiles >= 5 only; 0/1 = off): the K tiles are cut into `splitk` slices computed by different workgroups
   * into f32 partial sums in `ws` (>= splitk * batch * M * N * 4 bytes, 16-byte aligned, caller-owned), then a second
   * kernel sums the slices in a fixed order and applies the epilogue.  For small-M / huge-K problems (8x8 and 16x16
   * latent levels) that cannot fill 256 CUs with output tiles alone.  Needs N % 8 == 0, no GEGLU. */
  int splitk;
  void* ws;
  long long ws_bytes;
  /* GroupNorm statistics of the OUTPUT straight from the epilogue (GroupNorm32 util.py:191-193 follows almost every
   * convolution of the UNet: unet.py:149-153,173-180, attention.py:48-51): stats != NULL asks the launched kernel for the
   * per (row tile, column) statistics of the STORED 16-bit values, stats[tile_m][2][N] f32 (room for ceil(M / 64) * 2 * N
   * floats): [0] = their sum, [1] = M2 = the sum of their squared deviations from that tile-column's own mean (shifted
   * accumulation, pairwise merge: no cancellation when |mean| >> sigma).  IN/OUT: on return stats_rows = the rows per tile used (the launched kernel's tile
   * height; M % stats_rows == 0), or 0 when this launch could not produce them (GEGLU, transposed / f32 store, batch > 1, the persistent /
   * generic kernels, ragged M; split-K launches emit them from their reduce pass in 64-row tiles, tile 80 from its in-launch
   * reduction) — the caller then runs dbir_groupnorm_stats instead.
   * dbir_groupnorm_from_partials turns the sums of one or two column-adjacent producers into mean / variance. */
  float* stats;
  int stats_rows;
} dbir_gemm_desc;
int dbir_gemm(dbir_gemm_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * dbir_attention — flash-style softmax(Q K^T * scale) V, head_dim 64, no mask.
 * Replaces F.scaled_dot_product_attention at reference attention.py:206 (UNet/ControlNet self- and
 * cross-attention; xformers path attention.py:153 is the same math).
 *   Q : 16-bit, element (b, i, h, d) at Q[b*q_bstride + i*ldq + h*64 + d]
 *   K : same addressing with k_bstride/ldk over Lk keys
 *   Vt: TRANSPOSED values: element (b, h, d, j) at Vt[b*vt_bstride + (h*64 + d)*ldvt + j]; ldvt % 8 == 0 and
 *       columns j in [Lk, ldvt) must be readable (they are masked to zero in-kernel).
 *   O : element (b, i, h, d) at O[b*o_bstride + i*ldo + h*64 + d]
 */
int dbir_attention(int dtype, const void* Q, long long q_bstride, long long ldq, const void* K,
                   long long k_bstride, long long ldk, const void* Vt, long long vt_bstride, long long ldvt,
                   void* O, long long o_bstride, long long ldo, int B, int H, int Lq, int Lk, float scale,
                   void* stream);

/* dbir_window_attention — Swin (shifted-)window MHSA core: per window of ws*ws tokens
 * softmax(q k^T * scale + rel_pos_bias + shift_mask) v, including the cyclic roll and window
 * partition / reverse index maps.  Replaces reference swinir.py:126-149 together with 37-66, 222-243, 255-282.
 *   qkv : 16-bit [B, H, W, ld] with q at cols [0,C), k at [C,2C), v at [2C,3C) (real C = heads*hd)
 *   out : 16-bit [B, H, W, ldo], cols [0,C) written (token order = image order, roll undone)
 *   bias_table: f32 [(2ws-1)^2, heads]
 */
int dbir_window_attention(int dtype, const void* qkv, long long ld, void* out, long long ldo,
                          const float* bias_table, int B, int H, int W, int C, int heads, int ws, int shift,
                          float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Normalisation.  dbir_groupnorm replaces GroupNorm32 (+SiLU) reference util.py:191-193 / unet.py:149-153,
 * 173-180, nn.GroupNorm(32, C, eps=1e-6) at attention.py:48-51 and vae.py:18-21 (+swish vae.py:13-15).
 * x,y: NHWC 16-bit [B, HW, C] (row strides ldx/ldy); statistics in f32 over (HW, C/groups) per (b, group).
 * workspace: f32, at least B * (2*C*nchunk + 2*C) floats where nchunk = dbir_groupnorm_nchunk(HW, C).
 */
int dbir_groupnorm_nchunk(int HW, int C);
int dbir_groupnorm(int dtype, const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                   const float* beta, int B, int HW, int C, int groups, float eps, int silu, float* workspace,
                   void* stream);
/* Split form (tiled VAE, reference utils/tilevae/tilevae.py:232-304: GroupNorm statistics aggregated over the tiles of an
 * image): dbir_groupnorm_stats -> mean_var f32 [B][mean(groups) | biased var(groups)] of one tile (same workspace
 * contract as dbir_groupnorm); dbir_groupnorm_apply normalises with caller-supplied statistics of that layout. */
int dbir_groupnorm_stats(int dtype, const void* x, long long ldx, int B, int HW, int C, int groups, float* workspace,
                         float* mean_var, void* stream);
int dbir_groupnorm_apply(int dtype, const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                         const float* beta, const float* mean_var, int B, int HW, int C, int groups, float eps, int silu,
                         void* stream);
/* dbir_groupnorm_affine: statistics only, folded with gamma / beta into the per (sample, channel) affine map of the
 * normalisation, scale_shift f32 [B][2][C]: y = x * scale_shift[b][0][c] + scale_shift[b][1][c] — consumed by
 * dbir_xf_head, which applies it while it loads its activation panel (same workspace contract as dbir_groupnorm). */
int dbir_groupnorm_affine(int dtype, const void* x, long long ldx, int B, int HW, int C, int groups, float eps,
                          const float* gamma, const float* beta, float* workspace, float* scale_shift, void* stream);
/* dbir_groupnorm_from_partials: GroupNorm statistics from the column (sum, M2) pairs that dbir_gemm epilogues emit (dbir_gemm_desc.stats):
 * the normalised tensor [B, HW, C] is the column concatenation of producer 1 (N1 columns) and producer 2 (N2 columns, or
 * NULL / 0): p[tile][2][N] with `rows` rows per tile (HW % rows == 0).  Writes mean_var f32 [B][mean(groups) | biased
 * var(groups)] (the layout dbir_groupnorm_apply takes) and / or scale_shift f32 [B][2][C] (dbir_xf_head); either may be NULL. */
int dbir_groupnorm_from_partials(const float* p1, int N1, const float* p2, int N2, int rows, int B, int HW, int groups,
                                 float eps, const float* gamma, const float* beta, float* mean_var, float* scale_shift,
                                 void* stream);
/* dbir_groupnorm_apply_partials (round 5): dbir_groupnorm_from_partials + dbir_groupnorm_apply in one launch — every block
 * merges the (sum, M2) cells of its sample itself (same f64 merge), then normalises (+ SiLU) its rows of x [B, HW, N1 + N2]. */
int dbir_groupnorm_apply_partials(int dtype, const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                                  const float* beta, const float* p1, int N1, const float* p2, int N2, int rows, int B, int HW,
                                  int groups, float eps, int silu, void* stream);
/* dbir_layernorm: nn.LayerNorm rows (attention.py:255-257; swinir.py:205,211,764), eps 1e-5.
 * Normalises over the first C columns; columns [C, Cpad) of y are written as zero. */
int dbir_layernorm(int dtype, const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                   const float* beta, int rows, int C, int Cpad, float eps, void* stream);
/* dbir_softmax_rows: in-place row softmax over the first L columns of a 16-bit [rows, ld] matrix (f32 math),
 * columns [L, ld) set to zero.  Used by the single-head d=C VAE attention (vae.py:272). */
int dbir_softmax_rows(int dtype, void* x, long long ld, long long rows, int L, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused transformer-block kernels for the 64x64 latent level (C = 320) and the 32x32 level (C = 640), csrc/xformer.hip.
 * One SpatialTransformer
 * (reference attention.py:334-353 around BasicTransformerBlock._forward attention.py:265-274) becomes
 *   dbir_groupnorm_affine -> dbir_xf_head -> dbir_attention (self) -> dbir_xf_tail
 * instead of 16 launches; the activation is processed in 128-row (C = 320) / 64-row (C = 640) panels that stay in LDS
 * across the chained GEMMs / LayerNorms / text cross-attention / GEGLU feed-forward, only the weights stream in.
 * dbir_xf_geometry(C, &panel_rows, &head_tiles, &tail_tiles): panel height and stream lengths for C = 320 / 640
 * (dbir_xf_head_tiles() / dbir_xf_tail_tiles() are the C = 320 values).  At C = 640 a panel occupies a whole CU, so the
 * fused form pays from ~160 panels (M >= 10240 rows) on; below that launch the per-operator kernels.
 * Weights: ONE packed stream per kernel (diffbir_amd/xformer.py: tiles of dbir_xf_tile_bytes() bytes = 20 MFMA fragment
 * pieces of [32 rows][16 k] in LDS order + 512 B of f32 side data), dbir_xf_head_tiles() / dbir_xf_tail_tiles() tiles.
 * prm: f32 rows of C floats — head: proj_in bias, W beta1 for to_q / to_k / to_v; tail: attn1.to_out bias, Wq2 beta2,
 * attn2.to_out bias, ff.net.2 bias, proj_out bias (the LayerNorm affine maps are folded into the consuming weights).
 *
 * dbir_xf_head: x [M, C] (block input) -> h = proj_in(x * a + s) [M, C]; n = LayerNorm1(h); qk [M, 2C] = n Wq^T | n Wk^T;
 *   vt[b, c, l] = (n Wv^T)[b * L + l, c]  (attention.py:344-345, 266, 189-200 projections).  M = B * L, L % panel_rows == 0.
 * dbir_xf_tail: attn [Ms, C] (self-attention output), h [Ms, C], x [Ms, C] ->
 *   h1 = attn Wo1^T + b + h; a = softmax(LN2(h1) Wq^T K_ctx^T * scale) V_ctx; h2 = a Wo2^T + b + h1;
 *   h3 = (u * gelu(g)) W2^T + b + h2 with u | g = LN3(h2) W1^T + b; out [M, C] = h3 Wpo^T + b + x
 *   (attention.py:201-216, 266-273, 19-45, 350-353).  kfrag / vfrag: text-context K / V^T of the block, per (sample,
 *   head) in MFMA fragment order (xformer.py: pack_context_frags), Lk <= 96.
 *   pair_bs > 0: attn / h / x hold only the DISTINCT samples [G * pair_bs] of a classifier-free-guidance batch whose
 *   halves were identical so far (Ms = M / 2); output sample b reads source sample (b / (2 pair_bs)) * pair_bs + b % pair_bs.
 *   stop_after (tests only, 0 in production): dump an intermediate into `out` instead — 11: h1, 1: normalised h1 (no
 *   affine), 2: q, 3: cross-attention output, 14: h2, 4: normalised h2, 5: h3; 99 / 103 - 105: timing instantiations.
 *
 * Second generation (round 6, csrc/xformer2.hip): the same two entry points run kernels built around 8 waves x (64 rows x 80
 * columns) on v_mfma_f32_16x16x32 with the weights streamed straight into registers when the weight stream passed has the
 * second-generation length — dbir_xf2_geometry(C, &panel_rows, &head_bytes, &tail_bytes, &tail_prm_floats): per column group
 * (80 output columns) one flat sequence of 1 KB pieces (16 columns x 32 k, lane 16 lg + lr = W[n0 + lr][32 ks + 8 lg .. + 8]) in
 * consumption order + a copy of its first 10 pieces + 1 KB trailer (diffbir_amd/xformer.py: _pack_block_v2); tail prm = the 5
 * bias rows followed by the feed-forward projection bias table [20 chunks][C / 80][value | gate][16]; kfrag / vfrag in the
 * 16 x 16 fragment order (pack_context_frags, version 2).  Same arguments, same results (GELU by a sigmoid-form fit, max abs
 * error 8.1e-5); stop_after 99 = section timing in a DBIR_DIAG build. */
int dbir_xf2_geometry(int C, int* panel_rows, long long* head_bytes, long long* tail_bytes, int* tail_prm_floats);
int dbir_xf_tile_bytes(void);
int dbir_xf_head_tiles(void);
int dbir_xf_tail_tiles(void);
int dbir_xf_geometry(int C, int* panel_rows, int* head_tiles, int* tail_tiles);
int dbir_xf_head(int dtype, const void* x, long long ldx, const float* gn_scale_shift, void* h, long long ldh, void* qk,
                 long long ldqk, void* vt, long long vt_ld, long long vt_bstride, int M, int L, int C, const void* wstream,
                 long long wstream_bytes, const float* prm, void* stream);
int dbir_xf_tail(int dtype, const void* attn_out, long long ldo, const void* h, long long ldh, const void* x, long long ldx,
                 void* out, long long ldout, int M, int L, int C, int pair_bs, const void* wstream, long long wstream_bytes,
                 const float* prm, const void* kfrag, const void* vfrag, int Lk, float scale, int stop_after, void* stream);

/* ---------------------------------------------------------------------------------------------
 * OpenCLIP text tower (reference diffbir/model/clip.py:37-54 -> open_clip Transformer / ResidualAttentionBlock; its
 * nn.Linear layers are dbir_gemm calls).  csrc/clip.hip.
 * dbir_clip_embed: x[b, i, :] = token_embedding[tokens[b, i], :] + positional_embedding[i, :]  (f32 [B, L, W];
 *   tokens int64 [B, L], ids clamped to [0, vocab)).  Replaces clip.py:43 (`token_embedding(tokens) + positional_embedding`).
 * dbir_add_layernorm_f32: residual add + LayerNorm over f32 rows: x[rows, C] += y[rows, C] (y may be NULL) IN PLACE,
 *   then out = LN(x) * gamma + beta as 16-bit (out_f32 = 0: the operand of the next GEMM, row stride ldo) or f32
 *   (out_f32 = 1: ln_final).  Replaces `x = x + attn(ln_1(x))` / `x = x + mlp(ln_2(x))` / ln_final (clip.py:47-53) —
 *   the residual stream stays f32 as under the reference's autocast.
 * dbir_causal_attention: multi-head attention with the causal mask of the text tower (clip.py:46, attn_mask), head_dim
 *   64, L <= 128: qkv 16-bit [B, L, ld] with q at columns [0, H*64), k at [H*64, 2*H*64), v at [2*H*64, 3*H*64) (the
 *   in_proj layout of nn.MultiheadAttention); out 16-bit [B, L, ldo], columns [0, H*64).
 */
int dbir_clip_embed(const long long* tokens, const float* tok_emb, const float* pos, float* x, int B, int L, int W,
                    int vocab, void* stream);
int dbir_add_layernorm_f32(int dtype, float* x, const float* y, const float* gamma, const float* beta, void* out,
                           long long ldo, int out_f32, int rows, int C, float eps, void* stream);
int dbir_causal_attention(int dtype, const void* qkv, long long ld, void* out, long long ldo, int B, int H, int L,
                          float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layout / elementwise. */
/* out[m, :C] (ldo) = a[m, :C] (lda) + s * b[m, :C] (ldb); 16-bit; C % 8 == 0.
 * (`hs.pop() + control.pop()` and `h += control.pop()` controlnet.py:37,43; writes straight into the
 * concat buffer that replaces torch.cat controlnet.py:41-43.) */
int dbir_add_scaled(int dtype, const void* a, long long lda, const void* b, long long ldb, float s, void* out,
                    long long ldo, long long M, int C, void* stream);
/* 2x2 pixel-block <-> channel regrouping of NHWC 16-bit tensors (any 16-bit dtype; 16-byte moves):
 *   to_depth = 1: dst[b,i,j,(ky*2+kx)*C + c] = src[b,2i+ky,2j+kx,c]   ([B,2h,2w,C] -> [B,h,w,4C])
 *   to_depth = 0: dst[b,2i+ky,2j+kx,c] = src[b,i,j,(ky*2+kx)*C + c]   ([B,h,w,4C] -> [B,2h,2w,C])
 * h, w = the low-resolution extent.  Turns nn.Conv2d(k=2, s=2) / nn.ConvTranspose2d(k=2, s=2) of SCUNet
 * (scunet.py:179-212) into plain dbir_gemm calls. */
int dbir_block2x2(const void* src, long long lds, void* dst, long long ldd, int B, int h, int w, int C, int to_depth,
                  void* stream);
/* NCHW f32 sources -> NHWC 16-bit [B,H,W,Cpad]: channels = cat(src0[C0], src1[C1]) * scale + shift, zero
 * padded to Cpad.  (torch.cat((x, hint)) controlnet.py:317; `.type(self.dtype)` :320; `img*2-1` cldm.py:153.) */
int dbir_nchw_to_nhwc(int dtype, const float* src0, int C0, const float* src1, int C1, void* dst, int Cpad, int B,
                      int H, int W, float scale, float shift, void* stream);
/* NHWC (16-bit or f32 if src_f32) [B,H,W,ld] first C channels -> NCHW f32, y = x*scale + shift[c]. */
int dbir_nhwc_to_nchw(int dtype, const void* src, int src_f32, long long ld, float* dst, int C, int B, int H,
                      int W, float scale, const float* shift, void* stream);
/* SwinIR front end: (x - mean[c]) * range, PixelUnshuffle(r), NCHW f32 -> NHWC 16-bit [B,H/r,W/r,Cpad]
 * (swinir.py:860-861, 702-705). Output channel = c*r*r + dy*r + dx. */
int dbir_pixel_unshuffle(int dtype, const float* src, void* dst, int B, int C, int H, int W, int r, int Cpad,
                         const float* mean, float range, void* stream);
/* sinusoidal timestep embedding (util.py:128-148): t f32 [B] -> 16-bit [B, dim] = cat(cos, sin). */
int dbir_timestep_embedding(int dtype, const float* t, void* out, int B, int dim, float max_period, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sampler step math on f32 [B, n] latents (n = C*H*W), coefficients per batch row.
 * out = ca[b]*x + cb[b]*y + cc[b]*z + cd[b]*w   (any of y,z,w may be NULL)
 * Covers CFG mixing, x0-from-eps/v, posterior mean + sqrt(var)*noise (spaced_sampler.py:158,175-183) and the
 * DPM-Solver++ updates (dpm_solver_pytorch.py:451-459, 590-594, 842-849). */
int dbir_lincomb4(const float* x, const float* y, const float* z, const float* w, const float* ca,
                  const float* cb, const float* cc, const float* cd, float* out, int B, long long n, void* stream);
/* Fused spaced-DDPM step (spaced_sampler.py:144-184): model outputs oc/ou (f32 [B,n]), CFG scale s,
 * x0 = k_x[b]*x - k_o[b]*(ou + s*(oc-ou)); x_prev = c1[b]*x0 + c2[b]*x + sd[b]*noise. ou may be NULL (no CFG). */
int dbir_spaced_step(const float* x, const float* oc, const float* ou, const float* noise, float s,
                     const float* k_x, const float* k_o, const float* c1, const float* c2, const float* sd,
                     float* out, int B, long long n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Mixture-of-diffusers tiling (utils/common.py:172-232).
 * tile_gather: x f32 [B,C,H,W] -> tiles f32 [T*B, C, ts, ts] (tile-major: index t*B + b), coords int32 [T,2]
 * (hi, wi).  tile_accumulate: out[b,c,y,x] = sum_t tiles[t*B+b,c,y-hi_t,x-wi_t]*w[y-hi_t,x-wi_t] /
 * sum_t w[...] with tiles visited in increasing t (the reference's sequential order => same f32 rounding). */
int dbir_tile_gather(const float* x, float* tiles, const int* coords, int T, int B, int C, int H, int W, int ts,
                     void* stream);
int dbir_tile_accumulate(const float* tiles, const float* weights, const int* coords, float* out, int T, int B,
                         int C, int H, int W, int ts, void* stream);
/* Sharded form (one process per GPU): num = the un-normalised weighted sum over the SUBSET of tiles given
 * (tile-major [T*B,C,ts,ts] like above; tiles == NULL accumulates the weights alone = the normaliser [H,W] when called
 * with B = C = 1).  The per-rank partial sums are all-reduced by the host (RCCL), then
 * dbir_tile_normalize: out[bc, p] = num[bc, p] / den[p]. */
int dbir_tile_accumulate_partial(const float* tiles, const float* weights, const int* coords, float* num, int T,
                                 int B, int C, int H, int W, int ts, void* stream);
int dbir_tile_normalize(const float* num, const float* den, float* out, long long BC, long long HW, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pre/post image ops (pipeline.py:265-271, 306-320; utils/common.py:29-77). */
int dbir_u8_to_f32_nchw(const unsigned char* src, float* dst, int B, int H, int W, void* stream);
/* depthwise 3x3 [1,2,1]x[1,2,1]/16 blur with dilation `radius`, replicate padding; f32 planes [P,H,W]. */
int dbir_wavelet_blur(const float* src, float* dst, int P, int H, int W, int radius, void* stream);
/* out = (content - content_low) + style_low on f32 NCHW tensors of n elements (wavelet_reconstruction with the
 * telescoped sum of common.py:50-62: sum_i (img_i - low_i) = img_0 - low_last). */
int dbir_colorfix(const float* content, const float* content_low, const float* style_low, float* out,
                  long long n, void* stream);
/* dst_u8[b,y,x,c] = (uint8) clamp(src[b,c,y,x] * 255, 0, 255)  (pipeline.py:312-320; truncating cast). */
int dbir_f32_nchw_to_u8_nhwc(const float* src, unsigned char* dst, int B, int H, int W, void* stream);

/* dbir_copy_rows: dst[m, :C] (ldd) = src[m, :C] (lds) for 16-bit rows, C % 8 == 0, 16-byte aligned (the duplication of the
 * distinct samples of a classifier-free-guidance batch, model/unet.py `_expand_pairs`; keeps the whole network evaluation on
 * engine kernels so that it can be recorded into a dbir_plan). */
int dbir_copy_rows(const void* src, long long lds, void* dst, long long ldd, long long M, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Module-level entry point (`dbir_cldm_forward`): ONE host call per network evaluation.
 *
 * A dbir_plan is the recorded launch sequence of one ControlLDM evaluation (reference cldm.py:160-172: ControlNet + UNet,
 * ~600 calls of the operator entry points above on two streams) for fixed shapes, text context and control scales.  The host
 * side records it once while it runs the evaluation eagerly (diffbir_amd/plan.py: every argument of every call, which of
 * the plan's streams it went to, and the event record / wait pairs that order the two streams); dbir_plan_run replays it from
 * C — the same kernels on the same operands in the same per-stream order, no Python, no ctypes marshalling.  All device
 * memory the recorded calls touch is owned by the caller (a private PyTorch pool kept alive with the plan); the plan owns its
 * side streams, events and a copy of the call list.
 *   ops      : n_ops records.  fn >= 0: index of an operator entry point (dbir_plan_fn_index(name)); its arguments except the
 *              trailing stream in a[] (ints / long longs in .i, floats in .f, device pointers in .p); a dbir_gemm_desc
 *              argument is passed as a byte offset (.i) into `blob`.  fn == DBIR_PLAN_EVENT_RECORD: record event a[0].i on
 *              the op's stream; fn == DBIR_PLAN_STREAM_WAIT: the op's stream waits for event a[0].i.
 *   stream   : per op, a slot: 0 = the stream handed to dbir_plan_run, 1 .. n_streams - 1 = streams the plan creates.
 * dbir_plan_bind names the evaluation's static input / output buffers (slot 0 x, 1 t, 2 c_img, 3 eps output: device
 * pointers inside the recorded memory + byte sizes); dbir_cldm_forward copies the caller's x / t / c_img into them (skipped
 * when the pointers already match), replays the plan and copies the result to `eps` — everything enqueued on `stream`. */
typedef union dbir_arg { long long i; double f; void* p; } dbir_arg;
#define DBIR_PLAN_MAX_ARGS 24
#define DBIR_PLAN_EVENT_RECORD (-1)
#define DBIR_PLAN_STREAM_WAIT (-2)
typedef struct dbir_plan_op { int fn, stream, nargs, reserved; dbir_arg a[DBIR_PLAN_MAX_ARGS]; } dbir_plan_op;
typedef struct dbir_plan dbir_plan;
int dbir_plan_fn_index(const char* name);   /* -1: not a recordable entry point */
int dbir_plan_create(dbir_plan** out, const dbir_plan_op* ops, int n_ops, const void* blob, long long blob_bytes,
                     int n_streams, int n_events);
int dbir_plan_bind(dbir_plan* plan, int slot, void* device_ptr, long long bytes);
int dbir_plan_run(dbir_plan* plan, void* stream);
int dbir_plan_num_ops(const dbir_plan* plan);
int dbir_plan_destroy(dbir_plan* plan);
int dbir_cldm_forward(dbir_plan* plan, const float* x, const float* t, const float* c_img, float* eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DBIR_H */
