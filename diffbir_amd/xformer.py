"""Host side of the fused transformer-block kernels (csrc/xformer.hip): weight-stream packing and the op wrappers.

A SpatialTransformer of the 64x64 latent level (C = 320) or the 32x32 level (C = 640; reference attention.py:334-353
around BasicTransformerBlock._forward, attention.py:265-274) runs as

    groupnorm_affine -> xf_head -> attention (self) -> xf_tail

The kernels keep a 128-row (C = 320) / 64-row (C = 640) activation panel in LDS and stream ALL weights of the block through a 3-slot LDS ring as
one flat sequence of tiles; this module lays that sequence out.  A tile is 20 "fragment pieces" of 1 KB — piece
(block j, k-step s) holds rows [32 j, 32 j + 32) x columns [16 s, 16 s + 16) of a weight matrix [N, K] in the order an
MFMA 32x32x16 operand fragment is read: lane = 32 * hi + lq owns row 32 j + lq, columns 16 s + 8 hi .. + 8 — followed by
512 B of f32 side data (the GEGLU projection bias of a feed-forward chunk).  Tile order = consumption order:

The LayerNorm affine maps are folded into the GEMMs that consume the normalised rows — (xhat gamma + beta) W^T =
xhat (W diag(gamma))^T + W beta: to_q / to_k / to_v (norm1), attn2.to_q (norm2) and ff.net.0.proj (norm3) are stored
column-scaled, the W beta rows are the accumulators' initial values (parameter rows / the chunk's side data).

  (C = 320)
  head: proj_in, to_q, to_k, to_v                      10 tiles each: K tile kt = pieces [k-step 2 kt + ksl][block j]
  tail: attn1.to_out, attn2.to_q, attn2.to_out         10 tiles each, as above
        20 feed-forward chunks of 64 hidden units c:   4 tiles of ff.net.0.proj rows (value, gate blocks interleaved per
                                                       32: blocks 4c .. 4c+3), tile i = pieces [k-step 5 i + ksl][block];
                                                       2 tiles of ff.net.2 columns [64 c, 64 c + 64): [k-step 2 i + ksl][j]
        proj_out                                       10 tiles

  (C = 640: 20 column blocks, so a tile of an N = C GEMM is ONE k-step, 40 tiles per GEMM; a feed-forward chunk is 128
  hidden units = two runs of 8 GEGLU-projection tiles in the C = 320 format (run s = hidden blocks 4c + 2s, 4c + 2s + 1),
  then 8 one-k-step tiles of ff.net.2 columns [128 c, 128 c + 128).)  `geometry(C)` has the numbers.
"""
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import native

T = torch.Tensor
XC = 320
WIDTHS = (320, 640)
TILE_W, TILE_AUX = 20480, 512
TILE_BYTES = TILE_W + TILE_AUX
HEAD_TILES, TAIL_TILES = 40, 160            # C = 320
LK_PAD = 96
MIN_PANELS_640 = int(os.environ.get("DBIR_XF_MIN_PANELS_640", "128"))   # C = 640: 64-row panels, one per CU — below ~128 panels (half the CUs idle) the per-launch kernels win; 160 until round 6 (second-generation tail: 187 us fused against 240 per launch at 128 panels)
# Kernel generation the weights are packed for (the C side picks its kernels by the stream length):
#   2 (default, round 6) = csrc/xformer2.hip — 8 waves x (64 rows x 80 columns) on v_mfma_f32_16x16x32, weights streamed
#       straight into registers: per column group one flat sequence of 1 KB pieces (16 output columns x 32 k, lane
#       16 lg + lr = W[16 blk + lr, 32 ks + 8 lg .. + 8]) in consumption order, + a copy of its first 10 pieces (ring wrap);
#   1 = csrc/xformer.hip (rounds 3 - 5; kept for same-box A/B: DBIR_XF_VERSION=1).
XF_VERSION = int(os.environ.get("DBIR_XF_VERSION", "2"))
V2_RING = 10


@dataclass(frozen=True)
class Geometry:
    """csrc/xformer.hip XfCfg<C>."""
    C: int
    WN: int          # column groups of 160 (waves per row block)
    WM: int          # row blocks of 32
    BM: int          # panel rows
    NB: int          # column blocks of an N = C GEMM
    GNKS: int        # k-steps per tile of an N = C GEMM
    GNT: int         # tiles of a C x C GEMM
    NSUB: int        # GEGLU-projection runs per feed-forward chunk
    CHH: int         # hidden units per chunk
    CH: int          # chunks
    F1T: int         # tiles per GEGLU-projection run
    GKST: int        # k-steps of a chunk
    F2T: int         # tiles of a chunk's output projection
    head_tiles: int
    tail_tiles: int


def geometry(C: int) -> Geometry:
    assert C in WIDTHS, f"fused transformer kernels are built for C in {WIDTHS}"
    WN = C // 160
    WM = 8 // WN
    NB, KS = C // 32, C // 16
    GNKS = 20 // NB
    GNT = KS // GNKS
    NSUB, CHH = WN // 2, 32 * WN
    CH, F1T, GKST = 4 * C // CHH, KS // 5, CHH // 16
    F2T = GKST // GNKS
    return Geometry(C, WN, WM, 32 * WM, NB, GNKS, GNT, NSUB, CHH, CH, F1T, GKST, F2T, 4 * GNT,
                    4 * GNT + CH * (NSUB * F1T + F2T))


@dataclass
class XfBlock:
    """Packed weights of one transformer block for xf_head / xf_tail."""
    head_stream: T          # uint8 [HEAD_TILES, TILE_BYTES]
    head_prm: T             # f32 [4, C]: proj_in bias, then W beta1 for to_q / to_k / to_v (LayerNorm1 folded)
    tail_stream: T          # uint8 [TAIL_TILES, TILE_BYTES]
    tail_prm: T             # f32 [5, C]: to_out1 bias, Wq2 beta2, to_out2 bias, ff.net.2 bias, proj_out bias
    heads: int
    logical: Dict[str, T]   # the unpacked 16-bit weights / f32 vectors (validation tools and the CPU test double)
    version: int = 1        # kernel generation the streams are packed for


def _pieces(w: T) -> T:
    """[N, K] (N % 32 == 0, K % 16 == 0) -> [N/32, K/16, 64, 8]: fragment pieces (see module docstring)."""
    N, K = w.shape
    return w.reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(N // 32, K // 16, 64, 8).contiguous()


def _tiles_nc(w: T) -> T:
    """[C, K] -> [K / (16 GNKS) tiles, 20 pieces, 64, 8], piece index = ksl * (C / 32) + j."""
    p = _pieces(w)                                   # [C/32, K/16, 64, 8]
    nb = p.shape[0]
    gnks = 20 // nb
    nkt = p.shape[1] // gnks
    return p.reshape(nb, nkt, gnks, 64, 8).permute(1, 2, 0, 3, 4).reshape(nkt, 20, 64, 8)


def _geglu_interleave(w: T, b: T) -> Tuple[T, T]:
    """ff.net.0.proj [2*nh, K] (values | gates, attention.py:24-26) -> rows in blocks of 32: value block i, gate block i."""
    nh = w.shape[0] // 2
    wi = torch.stack([w[:nh].reshape(nh // 32, 32, -1), w[nh:].reshape(nh // 32, 32, -1)], dim=1).reshape(2 * nh, -1)
    bi = torch.stack([b[:nh].reshape(nh // 32, 32), b[nh:].reshape(nh // 32, 32)], dim=1).reshape(2 * nh)
    return wi, bi


def _finish_stream(tiles: T, aux: Optional[T], device) -> T:
    """tiles 16-bit [T, 20, 64, 8] (+ aux f32 [T, 128] or None) -> uint8 [T, TILE_BYTES] on `device`."""
    nt = tiles.shape[0]
    out = torch.zeros((nt, TILE_BYTES), dtype=torch.uint8)
    out[:, :TILE_W] = tiles.contiguous().view(torch.uint8).reshape(nt, TILE_W)
    if aux is not None:
        out[:, TILE_W:] = aux.contiguous().view(torch.uint8).reshape(nt, TILE_AUX)
    return out.to(device)


def _pieces16(w: T) -> T:
    """[N, K] (N % 16 == 0, K % 32 == 0) -> [N/16, K/32, 64, 8]: lane 16 lg + lr of piece (blk, ks) = W[16 blk + lr, 32 ks + 8 lg ..+8]."""
    N, K = w.shape
    return w.reshape(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(N // 16, K // 32, 64, 8).contiguous()


def _v2_gemm_pieces(w: T, cg: int) -> T:
    """[C, K] -> the pieces of column group cg (80 output columns) of a GEMM, k-step major: [K/32 * 5, 64, 8]."""
    p = _pieces16(w)[5 * cg:5 * cg + 5]              # [5, K/32, 64, 8]
    return p.permute(1, 0, 2, 3).reshape(-1, 64, 8)


def _v2_stream(groups, device) -> T:
    """per column group a list of piece tensors [n, 64, 8] (16 bit) -> uint8 [CG * (pieces + ring)] * 1024 on `device`."""
    out = []
    for parts in groups:
        st = torch.cat(parts)
        out.append(torch.cat([st, st[:V2_RING]]))
    body = torch.stack(out).contiguous().view(torch.uint8).reshape(-1)
    # + one trailing KB (never read): keeps every second-generation stream length distinct from the first generation's
    return torch.cat([body, torch.zeros(1024, dtype=torch.uint8)]).to(device)


def _pack_block_v2(g: Dict[str, T], dtype, device, head_prm: T, tail_prm5: T, q1w, k1w, v1w, q2w, ff1w, ff1b, logical) -> XfBlock:
    C = g["proj_in.w"].shape[0]
    CG, KS = C // 80, C // 32
    CHH = 16 * CG
    NCH, GK = 4 * C // CHH, CHH // 32
    h = lambda t: t.to(dtype)  # noqa: E731
    pi, pq, pk, pv = h(g["proj_in.w"]), h(q1w), h(k1w), h(v1w)
    o1, q2, o2, po = h(g["out1.w"]), h(q2w), h(g["out2.w"]), h(g["proj_out.w"])
    p1 = _pieces16(h(ff1w))                           # [8C/16, KS, 64, 8]: values then gates
    p2 = _pieces16(h(g["ff2.w"]))                     # [C/16, 4C/32, 64, 8]
    head_groups, tail_groups = [], []
    b1t = torch.zeros(NCH, CG, 2, 16)
    for cg in range(CG):
        head_groups.append([_v2_gemm_pieces(t, cg) for t in (pi, pq, pk, pv)])

        def f1(c):
            bv = (c * CHH + 16 * cg) // 16
            bg = (4 * C + c * CHH + 16 * cg) // 16
            return torch.stack([p1[bv], p1[bg]], dim=1).reshape(-1, 64, 8)      # [KS, 2] -> k-step major

        def f2(c):
            return p2[5 * cg:5 * cg + 5, c * GK:(c + 1) * GK].permute(1, 0, 2, 3).reshape(-1, 64, 8)

        ff = [f1(0)]
        for c in range(1, NCH):
            ff += [f1(c), f2(c - 1)]
        ff.append(f2(NCH - 1))
        tail_groups.append([_v2_gemm_pieces(o1, cg), _v2_gemm_pieces(q2, cg), _v2_gemm_pieces(o2, cg)] + ff + [_v2_gemm_pieces(po, cg)])
        for c in range(NCH):
            h0 = c * CHH + 16 * cg
            b1t[c, cg, 0] = ff1b[h0:h0 + 16]
            b1t[c, cg, 1] = ff1b[4 * C + h0:4 * C + h0 + 16]
    tail_prm = torch.cat([tail_prm5.reshape(-1).cpu(), b1t.reshape(-1)]).contiguous().to(device)
    return XfBlock(_v2_stream(head_groups, device), head_prm, _v2_stream(tail_groups, device), tail_prm, C // 64, logical, 2)


def pack_block(w: Dict[str, T], dtype, device, version: Optional[int] = None) -> XfBlock:
    """w: the block's tensors by short name (any float dtype / device) —
    proj_in.{w,b}, norm1.{w,b}, q1.w, k1.w, v1.w, out1.{w,b}, norm2.{w,b}, q2.w, out2.{w,b}, norm3.{w,b}, ff1.{w,b},
    ff2.{w,b}, proj_out.{w,b}."""
    version = XF_VERSION if version is None else version
    g = {k: v.detach().float().cpu().reshape(v.shape[0], -1) if v.dim() > 1 else v.detach().float().cpu() for k, v in w.items()}
    C = g["proj_in.w"].shape[0]
    geo = geometry(C)
    assert g["ff2.w"].shape == (C, 4 * C)
    # LayerNorm affine maps folded into the consuming GEMMs: (xhat * gamma + beta) W^T = xhat (W diag(gamma))^T + W beta
    fold = lambda wn, nn: (g[wn] * g[nn + ".w"][None, :], g[wn] @ g[nn + ".b"])  # noqa: E731
    (q1w, q1b), (k1w, k1b), (v1w, v1b) = fold("q1.w", "norm1"), fold("k1.w", "norm1"), fold("v1.w", "norm1")
    q2w, q2b = fold("q2.w", "norm2")
    ff1w, ff1b = fold("ff1.w", "norm3")
    ff1b = ff1b + g["ff1.b"]
    head_prm = torch.stack([g["proj_in.b"], q1b, k1b, v1b]).contiguous().to(device)
    tail_prm = torch.stack([g["out1.b"], q2b, g["out2.b"], g["ff2.b"], g["proj_out.b"]]).contiguous().to(device)
    logical = {k: (v.to(dtype) if k.endswith(".w") and not k.startswith("norm") else v).to(device) for k, v in g.items()}
    if torch.empty(0, dtype=dtype).element_size() != 2:  # f32 test double (CPU wiring tests): no kernel streams
        e = torch.empty((0, TILE_BYTES), dtype=torch.uint8, device=device)
        return XfBlock(e, head_prm, e, tail_prm, C // 64, logical)
    if version == 2:
        return _pack_block_v2(g, dtype, device, head_prm, tail_prm, q1w, k1w, v1w, q2w, ff1w, ff1b, logical)
    h = lambda name: g[name].to(dtype)  # noqa: E731  (weights are rounded to the 16-bit compute type once, here)
    head = torch.cat([_tiles_nc(t.to(dtype)) for t in (g["proj_in.w"], q1w, k1w, v1w)])
    w1, b1 = _geglu_interleave(ff1w, ff1b)
    p1 = _pieces(w1.to(dtype))                      # [8 C / 32 blocks, C / 16 k-steps, 64, 8]
    p2 = _pieces(h("ff2.w"))                        # [C / 32 blocks, 4 C / 16 k-steps, 64, 8]
    ff_tiles, ff_aux = [], []
    for c in range(geo.CH):
        for sub in range(geo.NSUB):                 # run = hidden blocks 2 sc, 2 sc + 1 (value / gate interleaved: 4 blocks)
            sc = c * geo.NSUB + sub
            for i in range(geo.F1T):                # piece = ksl * 4 + block
                ff_tiles.append(p1[4 * sc:4 * sc + 4, 5 * i:5 * i + 5].permute(1, 0, 2, 3).reshape(20, 64, 8))
                ff_aux.append(b1[128 * sc:128 * sc + 128])
        for i in range(geo.F2T):                    # piece = ksl * (C / 32) + j
            k0 = geo.GKST * c + geo.GNKS * i
            ff_tiles.append(p2[:, k0:k0 + geo.GNKS].permute(1, 0, 2, 3).reshape(20, 64, 8))
            ff_aux.append(torch.zeros(128))
    tail = torch.cat([_tiles_nc(h("out1.w")), _tiles_nc(q2w.to(dtype)), _tiles_nc(h("out2.w")),
                      torch.stack(ff_tiles), _tiles_nc(h("proj_out.w"))])
    tail_aux = torch.cat([torch.zeros(3 * geo.GNT, 128), torch.stack(ff_aux), torch.zeros(geo.GNT, 128)])
    assert head.shape[0] == geo.head_tiles and tail.shape[0] == geo.tail_tiles
    return XfBlock(_finish_stream(head, None, device), head_prm, _finish_stream(tail, tail_aux, device), tail_prm,
                   C // 64, logical)


def _pack_context_frags_v2(k: T, vt: T, Lk: int, heads: int) -> Tuple[T, T]:
    """xformer2.hip (16x16x32 MFMAs): kf [B, heads, 6, 2, 64, 8]: lane 16 lg + lr of (key block kb, d-step ds) =
    K[16 kb + lr, 64 h + 32 ds + 8 lg ..+8];  vf [B, heads, 4, 3, 64, 8]: lane 16 lg + lr of (d block db, key-step ss) =
    V^T[64 h + 16 db + lr, keys 32 ss + 4 lg + {0..3}, 32 ss + 16 + 4 lg + {0..3}] — the key order in which the softmax
    probabilities sit in the S^T accumulators.  Keys >= Lk are zero."""
    B, _, C = k.shape
    kp = torch.zeros((B, LK_PAD, C), dtype=k.dtype, device=k.device)
    kp[:, :Lk] = k[:, :Lk]
    kf = kp.reshape(B, 6, 16, heads, 2, 4, 8).permute(0, 3, 1, 4, 5, 2, 6).reshape(B, heads, 6, 2, 64, 8).contiguous()
    vp = torch.zeros((B, C, LK_PAD), dtype=vt.dtype, device=vt.device)
    vp[:, :, :Lk] = vt[:, :, :Lk]
    vv = vp.reshape(B, heads, 4, 16, 3, 2, 4, 4)                    # [B, h, db, lr, ss, half, lg, e]
    vf = vv.permute(0, 1, 2, 4, 6, 3, 5, 7).reshape(B, heads, 4, 3, 64, 8).contiguous()
    return kf, vf


def pack_context_frags(k: T, vt: T, Lk: int, heads: int, version: Optional[int] = None) -> Tuple[T, T]:
    """Text-context K [B, Lk, C] and V^T [B, C, >= Lk] of one block -> MFMA fragment order per (sample, head):
    kf [B, heads, 3, 4, 64, 8]: lane 32 hi + lq of (key block kb, d-step ks) = K[32 kb + lq, 64 h + 16 ks + 8 hi ..+8]
    vf [B, heads, 2, 6, 64, 8]: lane 32 hi + lq of (d block t, key-step s)  = V^T[64 h + 32 t + lq, keys 16 s + 4 hi +
    {0..3}, 16 s + 8 + 4 hi + {0..3}] — the key order in which the softmax probabilities sit in the S^T accumulators.
    Keys >= Lk are zero."""
    B, _, C = k.shape
    assert Lk <= LK_PAD and C == heads * 64
    if (XF_VERSION if version is None else version) == 2:
        return _pack_context_frags_v2(k, vt, Lk, heads)
    kp = torch.zeros((B, LK_PAD, C), dtype=k.dtype, device=k.device)
    kp[:, :Lk] = k[:, :Lk]
    kf = kp.reshape(B, 3, 32, heads, 4, 2, 8).permute(0, 3, 1, 4, 5, 2, 6).reshape(B, heads, 3, 4, 64, 8).contiguous()
    vp = torch.zeros((B, C, LK_PAD), dtype=vt.dtype, device=vt.device)
    vp[:, :, :Lk] = vt[:, :, :Lk]
    # key(s, hi, p) = 16 s + 8 (p >> 2) + 4 hi + (p & 3): split keys as [s 6][half 2][hi 2][e 4] -> order [s][hi][half][e]
    vv = vp.reshape(B, heads, 2, 32, 6, 2, 2, 4)                    # [B, h, t, lq, s, half, hi, e]
    vf = vv.permute(0, 1, 2, 4, 6, 3, 5, 7).reshape(B, heads, 2, 6, 64, 8).contiguous()
    return kf, vf


# ------------------------------------------------------------------------------------------------ op wrappers
def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dt(t: T) -> int:
    if t.dtype == torch.float16:
        return native.F16
    if t.dtype == torch.bfloat16:
        return native.BF16
    raise TypeError(f"expected a 16-bit tensor, got {t.dtype}")


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise native.NativeError("diffbir_amd ops need GPU tensors (no CPU fallback in the product path)")


def _rows_ld(t: T) -> Tuple[int, int]:
    """(rows, row stride) of a channels-last activation whose leading dims are dense over the row stride."""
    assert t.stride(-1) == 1
    ld = t.stride(-2)
    exp = ld
    for d in range(t.dim() - 2, 0, -1):
        exp *= t.shape[d]
        assert t.stride(d - 1) == exp or t.shape[d - 1] == 1, f"leading dims not dense: {t.shape} {t.stride()}"
    n = 1
    for s in t.shape[:-1]:
        n *= s
    return n, ld


def supported(C: int, L: int, Lk: int, M: Optional[int] = None) -> bool:
    """Can (and should) a block of inner width C with L rows per sample (M rows in all) run on the fused kernels?"""
    if C not in WIDTHS or not 0 < Lk <= LK_PAD:
        return False
    bm = geometry(C).BM
    if L % bm:
        return False
    if C == 640 and M is not None and M // bm < MIN_PANELS_640:
        return False
    # the C side addresses every tensor through a 2 GB buffer descriptor (the widest row is the [M, 2C] q | k output): a
    # larger problem takes the per-operator path instead of failing in dbir_xf_head / dbir_xf_tail
    if M is not None and (M * 2 * C) * 2 >= 0x7FFFFE00:
        return False
    return True


def groupnorm_affine(x: T, gamma: T, beta: T, eps: float, groups: int = 32) -> T:
    """x [B, HW.., C] 16-bit -> f32 [B, 2, C]: GroupNorm(x) = x * ab[b, 0, c] + ab[b, 1, c]."""
    _gpu(x, gamma, beta)
    B, C = x.shape[0], x.shape[-1]
    rows, ld = _rows_ld(x)
    HW = rows // B
    nchunk = native.lib().dbir_groupnorm_nchunk(HW, C)
    ws = torch.empty(B * (2 * C * nchunk + 2 * C), dtype=torch.float32, device=x.device)
    ab = torch.empty((B, 2, C), dtype=torch.float32, device=x.device)
    native.check(native.lib().dbir_groupnorm_affine(_dt(x), x.data_ptr(), ld, B, HW, C, groups, eps, gamma.data_ptr(),
                                                    beta.data_ptr(), ws.data_ptr(), ab.data_ptr(), _stream()),
                 "dbir_groupnorm_affine")
    return ab


def xf_head(x: T, ab: T, blk: XfBlock, L: int) -> Tuple[T, T, T]:
    """x [B, .., C] (M = B * L rows), ab f32 [B, 2, C] -> h [M, C], qk [B, L, 2C], vt [B, C, L]."""
    _gpu(x, ab)
    M, ldx = _rows_ld(x)
    B, C = M // L, x.shape[-1]
    h = torch.empty((M, C), dtype=x.dtype, device=x.device)
    qk = torch.empty((B, L, 2 * C), dtype=x.dtype, device=x.device)
    vt = torch.empty((B, C, L), dtype=x.dtype, device=x.device)
    assert ab.dtype == torch.float32 and ab.is_contiguous() and tuple(ab.shape) == (B, 2, C)
    native.check(native.lib().dbir_xf_head(_dt(x), x.data_ptr(), ldx, ab.data_ptr(), h.data_ptr(), C, qk.data_ptr(), 2 * C,
                                           vt.data_ptr(), L, C * L, M, L, C, blk.head_stream.data_ptr(),
                                           blk.head_stream.numel(), blk.head_prm.data_ptr(), _stream()), "dbir_xf_head")
    return h, qk, vt


def xf_tail(attn: T, h: T, x: T, blk: XfBlock, kf: T, vf: T, Lk: int, scale: float, L: int, out: Optional[T] = None,
            pair_bs: int = 0, stop_after: int = 0) -> T:
    """attn / h / x: [Ms, C] rows (x may be 4-D NHWC, any row stride); out: [M, C] rows (M = 2 Ms when pair_bs)."""
    _gpu(attn, h, x, out, kf, vf)
    Ms, ldo = _rows_ld(attn)
    _, ldh = _rows_ld(h)
    Mx, ldx = _rows_ld(x)
    C = attn.shape[-1]
    M = 2 * Ms if pair_bs else Ms
    assert Mx == Ms and _rows_ld(h)[0] == Ms
    if out is None:
        shp = (x.shape[0] * (2 if pair_bs else 1),) + tuple(x.shape[1:])
        out = torch.empty(shp, dtype=x.dtype, device=x.device)
    Mo, ldout = _rows_ld(out)
    assert Mo == M and kf.shape[0] == M // L and kf.is_contiguous() and vf.is_contiguous()
    native.check(native.lib().dbir_xf_tail(_dt(attn), attn.data_ptr(), ldo, h.data_ptr(), ldh, x.data_ptr(), ldx,
                                           out.data_ptr(), ldout, M, L, C, pair_bs, blk.tail_stream.data_ptr(),
                                           blk.tail_stream.numel(), blk.tail_prm.data_ptr(), kf.data_ptr(), vf.data_ptr(),
                                           Lk, scale, stop_after, _stream()), "dbir_xf_tail")
    return out
