"""CPU checks of the fused transformer-block path (csrc/xformer.hip, diffbir_amd/xformer.py):

  * the weight-stream packer against an index-level restatement of what the kernel reads (tile / piece / lane / element
    addressing of every phase, the GEGLU side data, the text-context fragment order, the LDS operand image offsets);
  * the host wiring (`_DiffusionNet._attn` fused branch, incl. the shared CFG prefix) on the PyTorch test double against
    the per-launch path it replaces.
The kernels themselves are compared with the f32 statement in tests/test_kernels_gpu.py (-m gpu)."""
import numpy as np
import pytest
import torch

from diffbir_amd import configs, xformer
from diffbir_amd.utils.synth import synth_state_dict
from tests import emu_ops

def _rand_block(C, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * scale  # noqa: E731
    w = {}
    for n in ("proj_in", "out1", "out2", "proj_out"):
        w[n + ".w"], w[n + ".b"] = r(C, C) / C ** 0.5, r(C) * 0.1
    for n in ("q1", "k1", "v1", "q2"):
        w[n + ".w"] = r(C, C) / C ** 0.5
    for n in ("norm1", "norm2", "norm3"):
        w[n + ".w"], w[n + ".b"] = 1 + 0.1 * r(C), 0.1 * r(C)
    w["ff1.w"], w["ff1.b"] = r(8 * C, C) / C ** 0.5, r(8 * C) * 0.1
    w["ff2.w"], w["ff2.b"] = r(C, 4 * C) / (4 * C) ** 0.5, r(C) * 0.1
    return w


def _tile(stream, t):
    raw = stream[t].numpy()
    pieces = raw[:xformer.TILE_W].view(np.float16).reshape(20, 64, 8)
    aux = raw[xformer.TILE_W:].view(np.float32)
    return pieces, aux


# What the kernels read, restated from csrc/xformer.hip with its own constants (NOT through xformer.geometry):
#   wave (wm, wn) owns rows [32 wm, +32) x columns [160 wn, +160) = blocks 5 wn .. 5 wn + 4;  WN = C / 160
#   XF_GEMMCC: GNT tiles of GNKS k-steps; W fragment of block j, k-step ks of tile i: piece ks * (C / 32) + j
#   GEGLU run (sub s of chunk c): F1T tiles of 5 k-steps, piece ks * 4 + 2 (wn & 1) + {0: value, 1: gate}, for the waves
#     with wn / 2 == s; hidden block of wave wn in chunk c = (C / 160) c + wn
#   feed-forward output: F2T tiles of GNKS k-steps of the chunk's GKST = 2 WN k-steps, piece ks * (C / 32) + j
KCONST = {320: dict(WN=2, GNKS=2, GNT=10, NSUB=1, F1T=4, GKST=4, F2T=2, HEAD=40, TAIL=160),
          640: dict(WN=4, GNKS=1, GNT=40, NSUB=2, F1T=8, GKST=8, F2T=8, HEAD=160, TAIL=640)}


def _read_gemm(stream, t0, C):
    """What XF_GEMMCC multiplies: W[n][k] for n < C, k < C from tiles t0 .. t0 + GNT - 1."""
    k = KCONST[C]
    W = np.zeros((C, C), np.float16)
    for kt in range(k["GNT"]):
        pieces, _ = _tile(stream, t0 + kt)
        for ksl in range(k["GNKS"]):
            for wn in range(k["WN"]):
                for jl in range(5):
                    frag = pieces[ksl * (C // 32) + 5 * wn + jl]       # tbase + (ks * PSTR + WFIRST + jl) * 1024
                    for lane in range(64):
                        lq, hi = lane & 31, lane >> 5
                        W[32 * (5 * wn + jl) + lq, 16 * (k["GNKS"] * kt + ksl) + 8 * hi:][:8] = frag[lane]
    return W


@pytest.mark.parametrize("C", [320, 640])
def test_weight_stream_is_what_the_kernel_reads(C):
    k = KCONST[C]
    w = _rand_block(C)
    blk = xformer.pack_block(w, torch.float16, torch.device("cpu"), version=1)
    assert tuple(blk.head_stream.shape) == (k["HEAD"], xformer.TILE_BYTES)
    assert tuple(blk.tail_stream.shape) == (k["TAIL"], xformer.TILE_BYTES)
    geo = xformer.geometry(C)
    assert (geo.WN, geo.GNKS, geo.GNT, geo.NSUB, geo.F1T, geo.GKST, geo.F2T, geo.head_tiles, geo.tail_tiles) == \
        tuple(k[n] for n in ("WN", "GNKS", "GNT", "NSUB", "F1T", "GKST", "F2T", "HEAD", "TAIL"))
    assert geo.BM * C * 2 == 81920 and geo.CH == 20
    h16 = lambda n: w[n].half().numpy()  # noqa: E731
    # LayerNorm affine maps are folded into the consuming weights: W diag(gamma), bias W beta
    fw = lambda n, ln: (w[n] * w[ln + ".w"][None, :]).half().numpy()  # noqa: E731
    fb = lambda n, ln: (w[n] @ w[ln + ".b"]).numpy()                   # noqa: E731
    G = k["GNT"]
    for i, exp in enumerate((h16("proj_in.w"), fw("q1.w", "norm1"), fw("k1.w", "norm1"), fw("v1.w", "norm1"))):
        assert np.array_equal(_read_gemm(blk.head_stream, G * i, C), exp), i
    per_chunk = k["NSUB"] * k["F1T"] + k["F2T"]
    for t0, exp in ((0, h16("out1.w")), (G, fw("q2.w", "norm2")), (2 * G, h16("out2.w")),
                    (3 * G + 20 * per_chunk, h16("proj_out.w"))):
        assert np.array_equal(_read_gemm(blk.tail_stream, t0, C), exp), t0
    # feed-forward: chunk c = hidden units [32 WN c, +32 WN); wave column group wn owns hidden block WN c + wn
    w1, b1, w2 = fw("ff1.w", "norm3"), (w["ff1.b"] + w["ff1.w"] @ w["norm3.b"]).numpy(), h16("ff2.w")
    WN = k["WN"]
    for c in (0, 7, 19):
        t0 = 3 * G + per_chunk * c
        val = np.zeros((WN, 32, C), np.float16)
        gate = np.zeros((WN, 32, C), np.float16)
        for sub in range(k["NSUB"]):
            for i in range(k["F1T"]):                        # XF_RUN_BODY(F1T, 5, 2, 4, ..., 2 * (wn & 1), gacc)
                pieces, aux = _tile(blk.tail_stream, t0 + sub * k["F1T"] + i)
                for ksl in range(5):
                    for wl in range(2):
                        wn = 2 * sub + wl
                        for blkid, dst in ((0, val), (1, gate)):
                            frag = pieces[ksl * 4 + 2 * wl + blkid]
                            for lane in range(64):
                                lq, hi = lane & 31, lane >> 5
                                dst[wn, lq, 16 * (5 * i + ksl) + 8 * hi:][:8] = frag[lane]
                if i == 0:                                   # bias: aux[wl * 64 + blk * 32 + col]
                    for wl in range(2):
                        hb = WN * c + 2 * sub + wl
                        assert np.array_equal(aux[wl * 64:wl * 64 + 32], b1[32 * hb:32 * hb + 32])
                        assert np.array_equal(aux[wl * 64 + 32:wl * 64 + 64], b1[4 * C + 32 * hb:4 * C + 32 * hb + 32])
        for wn in range(WN):
            hb = WN * c + wn
            assert np.array_equal(val[wn], w1[32 * hb:32 * hb + 32])
            assert np.array_equal(gate[wn], w1[4 * C + 32 * hb:4 * C + 32 * hb + 32])
        got = np.zeros((C, 32 * WN), np.float16)
        for i in range(k["F2T"]):                            # XF_RUN_BODY(F2T, GNKS, 5, C / 32, GB + (wm * GKST + GNKS i + ks), 5 wn)
            pieces, _ = _tile(blk.tail_stream, t0 + k["NSUB"] * k["F1T"] + i)
            for ksl in range(k["GNKS"]):
                for j in range(C // 32):
                    frag = pieces[ksl * (C // 32) + j]
                    for lane in range(64):
                        lq, hi = lane & 31, lane >> 5
                        got[32 * j + lq, 16 * (k["GNKS"] * i + ksl) + 8 * hi:][:8] = frag[lane]
        assert np.array_equal(got, w2[:, 32 * WN * c:32 * WN * c + 32 * WN])
    prm = blk.tail_prm.numpy()
    for row, exp in enumerate((w["out1.b"].numpy(), fb("q2.w", "norm2"), w["out2.b"].numpy(), w["ff2.b"].numpy(),
                               w["proj_out.b"].numpy())):
        assert np.array_equal(prm[row], exp), row
    assert np.array_equal(blk.head_prm.numpy(), np.stack([w["proj_in.b"].numpy(), fb("q1.w", "norm1"), fb("k1.w", "norm1"),
                                                          fb("v1.w", "norm1")]))


def test_dispatch_rule():
    """Which blocks run on the fused kernels: C = 320 for any whole-panel sequence length; C = 640 only from ~160 panels of
    64 rows on (one panel occupies a CU: fewer would leave most of the chip idle); text context <= 96 tokens."""
    assert xformer.supported(320, 4096, 77) and xformer.supported(320, 128, 77, 128)
    assert not xformer.supported(320, 192, 77) and not xformer.supported(320, 4096, 97) and not xformer.supported(1280, 256, 77)
    assert xformer.supported(640, 1024, 77) and xformer.supported(640, 1024, 77, 16 * 1024)
    assert xformer.supported(640, 1024, 77, xformer.MIN_PANELS_640 * 64)
    assert not xformer.supported(640, 1024, 77, xformer.MIN_PANELS_640 * 64 - 64)
    assert not xformer.supported(640, 96, 77) and not xformer.supported(640, 1024, 0)


def _xoff(rowblk, kst, col, lq):  # xformer.hip: xoff()
    return ((rowblk * kst + (col >> 4)) * 2 + ((col >> 3) & 1)) * 512 + lq * 16 + ((col >> 2) & 1) * 8


def test_operand_image_offsets_are_consistent():
    """An accumulator quad written at xoff() is read back by the A-side fragment read `(rowblk * kst + kstep) * 1024 +
    lane * 16` as columns 16 kstep + 8 hi + e of row lq."""
    for kst, nrb in ((20, 4), (4, 4), (40, 2), (8, 2)):   # X / chunk image at C = 320, at C = 640
        seen = set()
        for rb in range(nrb):
            for col in range(0, 16 * kst, 4):
                for lq in range(32):
                    off = _xoff(rb, kst, col, lq)
                    kstep, hi, e0 = col // 16, (col // 8) & 1, col % 8
                    assert off == (rb * kst + kstep) * 1024 + (hi * 32 + lq) * 16 + e0 * 2
                    assert off % 8 == 0 and off not in seen
                    seen.add(off)
        assert len(seen) * 8 == nrb * 32 * 16 * kst * 2  # the image is covered exactly once


@pytest.mark.parametrize("C", [320, 640])
def test_context_fragments_are_what_the_kernel_reads(C):
    B, heads, Lk = 2, C // 64, 77
    g = torch.Generator().manual_seed(3)
    k = torch.randn(B, Lk, C, generator=g).half()
    vt = torch.randn(B, C, 80, generator=g).half()
    kf, vf = xformer.pack_context_frags(k, vt, Lk, heads, version=1)
    assert tuple(kf.shape) == (B, heads, 3, 4, 64, 8) and tuple(vf.shape) == (B, heads, 2, 6, 64, 8)
    kn, vn, kfn, vfn = k.numpy(), vt.numpy(), kf.numpy(), vf.numpy()
    for b in range(B):
        for h in range(heads):
            for lane in range(64):
                lq, hi = lane & 31, lane >> 5
                for kb in range(3):       # S^T block kb: A operand row = key 32 kb + lq, k = d 16 ks + 8 hi + e
                    for ks in range(4):
                        key = 32 * kb + lq
                        exp = kn[b, key, 64 * h + 16 * ks + 8 * hi:][:8] if key < Lk else np.zeros(8, np.float16)
                        assert np.array_equal(kfn[b, h, kb, ks, lane], exp)
                for t in range(2):        # O^T block t: A operand row = d 32 t + lq; key of element p of key-step s
                    for s in range(6):
                        for p in range(8):
                            key = 16 * s + 4 * hi + (p & 3) + 8 * (p >> 2)     # = where P sits in the S^T accumulators
                            exp = vn[b, 64 * h + 32 * t + lq, key] if key < Lk else np.float16(0)
                            assert vfn[b, h, t, s, lane, p] == exp


# ---------------------------------------------------------------------------------------------------------------------
# second-generation kernels (csrc/xformer2.hip), restated from the kernel source with its own constants:
#   wave = (row group rg, column group cg): rows [64 rg, +64) x columns [80 cg, +80) as 4 x 5 blocks of 16 x 16; CG = C / 80
#   a column group's stream = 1 KB pieces in consumption order; piece (16 columns n0.., k-step ks of 32): lane 16 lg + lr holds
#   W[n0 + lr][32 ks + 8 lg .. + 8]
#   gemm5: for ks: pieces j = 0..4 = columns 80 cg + 16 j
#   feed-forward order: F1(0), [F1(c), F2(c - 1)] c = 1 .. 19, F2(19); F1(c): for ks: (value, gate) of hidden units
#   16 CG c + 16 cg ..+16; F2(c): for k in range(GK = CG / 2): pieces j = 0..4, k-step (16 CG c) / 32 + k of ff.net.2
#   stream = out1, q2, out2, feed-forward, proj_out (tail) / proj_in, q, k, v (head), + a copy of the first 10 pieces, + 1 KB trailer
def _v2_streams(raw, CG, pieces):
    body = raw.numpy()[:-1024].view(np.float16).reshape(CG, pieces + xformer.V2_RING, 64, 8)
    for cg in range(CG):
        assert np.array_equal(body[cg, pieces:], body[cg, :xformer.V2_RING])   # ring wrap copy
    return body


def _v2_read_gemm(st, p0, C):
    """W[n][k] as gemm5 multiplies it, from pieces p0 .. of every column group."""
    CG, KS = C // 80, C // 32
    W = np.zeros((C, C), np.float16)
    for cg in range(CG):
        for ks in range(KS):
            for j in range(5):
                frag = st[cg, p0 + 5 * ks + j]
                for lane in range(64):
                    lr, lg = lane & 15, lane >> 4
                    W[80 * cg + 16 * j + lr, 32 * ks + 8 * lg:][:8] = frag[lane]
    return W


@pytest.mark.parametrize("C", [320, 640])
def test_weight_stream_v2_is_what_the_kernel_reads(C):
    CG, KS = C // 80, C // 32
    NCH, GK, GP = 20, CG // 2, 5 * (C // 32)
    w = _rand_block(C, seed=2)
    blk = xformer.pack_block(w, torch.float16, torch.device("cpu"), version=2)
    assert blk.version == 2
    head_pieces, tail_pieces = 4 * GP, 4 * GP + NCH * (2 * KS + 5 * GK)
    assert blk.head_stream.numel() == CG * (head_pieces + 10) * 1024 + 1024
    assert blk.tail_stream.numel() == CG * (tail_pieces + 10) * 1024 + 1024
    # the two generations are told apart by their stream lengths (csrc/xformer.hip dispatch): never equal
    v1 = xformer.pack_block(w, torch.float16, torch.device("cpu"), version=1)
    assert v1.head_stream.numel() != blk.head_stream.numel() and v1.tail_stream.numel() != blk.tail_stream.numel()
    hs, ts = _v2_streams(blk.head_stream, CG, head_pieces), _v2_streams(blk.tail_stream, CG, tail_pieces)
    h16 = lambda n: w[n].half().numpy()  # noqa: E731
    fw = lambda n, ln: (w[n] * w[ln + ".w"][None, :]).half().numpy()  # noqa: E731
    fb = lambda n, ln: (w[n] @ w[ln + ".b"]).numpy()                   # noqa: E731
    for i, exp in enumerate((h16("proj_in.w"), fw("q1.w", "norm1"), fw("k1.w", "norm1"), fw("v1.w", "norm1"))):
        assert np.array_equal(_v2_read_gemm(hs, GP * i, C), exp), i
    ff0 = 3 * GP
    for p0, exp in ((0, h16("out1.w")), (GP, fw("q2.w", "norm2")), (2 * GP, h16("out2.w")),
                    (ff0 + NCH * (2 * KS + 5 * GK), h16("proj_out.w"))):
        assert np.array_equal(_v2_read_gemm(ts, p0, C), exp), p0
    w1, b1, w2 = fw("ff1.w", "norm3"), (w["ff1.b"] + w["ff1.w"] @ w["norm3.b"]).numpy(), h16("ff2.w")
    F1P, F2P = 2 * KS, 5 * GK

    def f1_pos(c):   # F1(0) F1(1) F2(0) F1(2) F2(1) ...
        return ff0 + (0 if c == 0 else F1P + (c - 1) * (F1P + F2P))

    def f2_pos(c):
        return ff0 + (NCH * F1P + (NCH - 1) * F2P if c == NCH - 1 else 2 * F1P + c * (F1P + F2P))

    prm = blk.tail_prm.numpy()
    b1t = prm[5 * C:].reshape(NCH, CG, 2, 16)
    for c in (0, 1, 7, 19):
        for cg in range(CG):
            h0 = 16 * CG * c + 16 * cg
            val, gate = np.zeros((16, C), np.float16), np.zeros((16, C), np.float16)
            for ks in range(KS):
                for nb, dst in ((0, val), (1, gate)):
                    frag = ts[cg, f1_pos(c) + 2 * ks + nb]
                    for lane in range(64):
                        lr, lg = lane & 15, lane >> 4
                        dst[lr, 32 * ks + 8 * lg:][:8] = frag[lane]
            assert np.array_equal(val, w1[h0:h0 + 16]) and np.array_equal(gate, w1[4 * C + h0:4 * C + h0 + 16])
            assert np.array_equal(b1t[c, cg, 0], b1[h0:h0 + 16]) and np.array_equal(b1t[c, cg, 1], b1[4 * C + h0:4 * C + h0 + 16])
            got = np.zeros((80, 16 * CG), np.float16)
            for k in range(GK):
                for j in range(5):
                    frag = ts[cg, f2_pos(c) + 5 * k + j]
                    for lane in range(64):
                        lr, lg = lane & 15, lane >> 4
                        got[16 * j + lr, 32 * k + 8 * lg:][:8] = frag[lane]
            assert np.array_equal(got, w2[80 * cg:80 * cg + 80, 16 * CG * c:16 * CG * (c + 1)])
    for row, exp in enumerate((w["out1.b"].numpy(), fb("q2.w", "norm2"), w["out2.b"].numpy(), w["ff2.b"].numpy(),
                               w["proj_out.b"].numpy())):
        assert np.array_equal(prm[row * C:(row + 1) * C], exp), row


def test_operand_image_offsets_v2_are_consistent():
    """xformer2.hip: the 4 consecutive columns a lane holds of 16-column block t of row block rbg are written at
    rbg * KST * 1024 + t * 512 + lanew and read back by the fragment read (rbg * KST + ks) * 1024 + lane * 16 as columns
    32 ks + 8 lg + e of row lr."""
    for kst, nrb in ((10, 8), (2, 8), (20, 4), (4, 4)):   # X / chunk image at C = 320, at C = 640
        seen = set()
        for rbg in range(nrb):
            for t in range(2 * kst):
                for lane in range(64):
                    lr, lg = lane & 15, lane >> 4
                    off = rbg * kst * 1024 + t * 512 + (lg >> 1) * 256 + lr * 16 + (lg & 1) * 8
                    col = 16 * t + 4 * lg                  # first of the lane's 4 columns
                    ks, lgr, e0 = col // 32, (col % 32) // 8, col % 8
                    assert off == (rbg * kst + ks) * 1024 + (lgr * 16 + lr) * 16 + e0 * 2
                    assert off not in seen
                    seen.add(off)
        assert len(seen) * 8 == nrb * 16 * 32 * kst * 2


@pytest.mark.parametrize("C", [320, 640])
def test_context_fragments_v2_are_what_the_kernel_reads(C):
    B, heads, Lk = 2, C // 64, 77
    g = torch.Generator().manual_seed(3)
    k = torch.randn(B, Lk, C, generator=g).half()
    vt = torch.randn(B, C, 80, generator=g).half()
    kf, vf = xformer.pack_context_frags(k, vt, Lk, heads, version=2)
    assert tuple(kf.shape) == (B, heads, 6, 2, 64, 8) and tuple(vf.shape) == (B, heads, 4, 3, 64, 8)
    kn, vn, kfn, vfn = k.numpy(), vt.numpy(), kf.numpy(), vf.numpy()
    for b in range(B):
        for h in range(heads):
            for lane in range(64):
                lr, lg = lane & 15, lane >> 4
                for kb in range(6):       # S^T block kb: A operand row = key 16 kb + lr, k = d 32 ds + 8 lg + e
                    for ds in range(2):
                        key = 16 * kb + lr
                        exp = kn[b, key, 64 * h + 32 * ds + 8 * lg:][:8] if key < Lk else np.zeros(8, np.float16)
                        assert np.array_equal(kfn[b, h, kb, ds, lane], exp)
                for db in range(4):       # O^T block db: A operand row = d 16 db + lr; element i of key-step ss = the key whose
                    for ss in range(3):   # probability sits in element i of the lane's P fragment: blocks 2 ss, 2 ss + 1, keys 4 lg + e
                        for i in range(8):
                            key = 32 * ss + 16 * (i >> 2) + 4 * lg + (i & 3)
                            exp = vn[b, 64 * h + 16 * db + lr, key] if key < Lk else np.float16(0)
                            assert vfn[b, h, db, ss, lane, i] == exp


def _small_unet(ch=320):
    from diffbir_amd.model.unet import ControlledUnetModel
    cfg = configs._unet(ch, 64, mult=(1,), attn=(1,), nrb=1)
    net = ControlledUnetModel(**cfg)
    net.load_state_dict(synth_state_dict(net._spec, seed=5), strict=True)
    net._dtype, net._packed = torch.float32, False
    return net


@torch.no_grad()
@pytest.mark.parametrize("pair,ch", [(None, 320), ((1, 1), 320), (None, 640)])
def test_fused_block_wiring_equals_per_launch_path(monkeypatch, pair, ch):
    """groupnorm_affine -> xf_head -> attention -> xf_tail (test double, f32) == the 16-launch path, with and without
    the shared CFG prefix (pair_bs source-row mapping, full-batch text context)."""
    emu_ops.install(monkeypatch)
    from diffbir_amd.model import unet as unet_mod
    assert unet_mod.FUSED_XF and ch in unet_mod.FUSED_XF_WIDTHS
    monkeypatch.setattr(xformer, "MIN_PANELS_640", 1)     # (the C = 640 kernels are dispatched from ~160 panels on)
    net = _small_unet(ch)
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(1, 4, 16, 8, generator=g)           # latent 16 x 8 -> L = 128 rows per sample
    x = torch.cat([x1, x1]) if pair else torch.randn(2, 4, 16, 8, generator=g)
    t = torch.tensor([500.0, 500.0]) if pair else torch.tensor([500.0, 20.0])
    ctx = torch.randn(2, 77, 64, generator=g)
    fused = net(x, t, ctx, pair=pair)
    assert all(a.xf is not None for a in net._attn_layers) and len(net._attn_layers) == 4
    for a in net._attn_layers:                            # same packed network, per-launch transformer blocks
        a.xf = None
    net._ctx_cache.clear()
    plain = net(x, t, ctx, pair=pair)
    err = ((fused - plain).norm() / plain.norm()).item()
    assert err < 2e-5, err
    assert (fused[0] - fused[1]).abs().max() > 1e-3      # the two contexts really differ
