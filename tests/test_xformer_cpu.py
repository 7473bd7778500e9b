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

C = 320


def _rand_block(seed=0, scale=1.0):
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


def _read_gemm320(stream, t0):
    """What the kernel's XF_GEMM320 multiplies: W[n][k] for n < 320, k < 320 from tiles t0 .. t0 + 9."""
    W = np.zeros((C, C), np.float16)
    for kt in range(10):
        pieces, _ = _tile(stream, t0 + kt)
        for ksl in range(2):
            for wn in range(2):
                for jl in range(5):
                    frag = pieces[ksl * 10 + 5 * wn + jl]            # tbase + (ks * PSTR + WFIRST + jl) * 1024
                    for lane in range(64):
                        lq, hi = lane & 31, lane >> 5
                        W[32 * (5 * wn + jl) + lq, 32 * kt + 16 * ksl + 8 * hi:][:8] = frag[lane]
    return W


def test_weight_stream_is_what_the_kernel_reads():
    w = _rand_block()
    blk = xformer.pack_block(w, torch.float16, torch.device("cpu"))
    assert tuple(blk.head_stream.shape) == (xformer.HEAD_TILES, xformer.TILE_BYTES)
    assert tuple(blk.tail_stream.shape) == (xformer.TAIL_TILES, xformer.TILE_BYTES)
    h16 = lambda n: w[n].half().numpy()  # noqa: E731
    # LayerNorm affine maps are folded into the consuming weights: W diag(gamma), bias W beta
    fw = lambda n, ln: (w[n] * w[ln + ".w"][None, :]).half().numpy()  # noqa: E731
    fb = lambda n, ln: (w[n] @ w[ln + ".b"]).numpy()                   # noqa: E731
    for i, exp in enumerate((h16("proj_in.w"), fw("q1.w", "norm1"), fw("k1.w", "norm1"), fw("v1.w", "norm1"))):
        assert np.array_equal(_read_gemm320(blk.head_stream, 10 * i), exp), i
    for t0, exp in ((0, h16("out1.w")), (10, fw("q2.w", "norm2")), (20, h16("out2.w")), (150, h16("proj_out.w"))):
        assert np.array_equal(_read_gemm320(blk.tail_stream, t0), exp), t0
    # feed-forward: chunk c = hidden units [64 c, 64 c + 64); wave column half wn owns hidden block 2 c + wn
    w1, b1, w2 = fw("ff1.w", "norm3"), (w["ff1.b"] + w["ff1.w"] @ w["norm3.b"]).numpy(), h16("ff2.w")
    for c in (0, 7, 19):
        t0 = 30 + 6 * c
        val = np.zeros((2, 32, C), np.float16)
        gate = np.zeros((2, 32, C), np.float16)
        for i in range(4):                                   # XF_TILE(5, 2, 4, ..., 2 * wn, gacc)
            pieces, aux = _tile(blk.tail_stream, t0 + i)
            for ksl in range(5):
                for wn in range(2):
                    for blkid, dst in ((0, val), (1, gate)):
                        frag = pieces[ksl * 4 + 2 * wn + blkid]
                        for lane in range(64):
                            lq, hi = lane & 31, lane >> 5
                            dst[wn, lq, 16 * (5 * i + ksl) + 8 * hi:][:8] = frag[lane]
            if i == 0:                                       # bias: aux[wn * 64 + blk * 32 + col]
                for wn in range(2):
                    hb = 2 * c + wn
                    assert np.array_equal(aux[wn * 64:wn * 64 + 32], b1[32 * hb:32 * hb + 32])
                    assert np.array_equal(aux[wn * 64 + 32:wn * 64 + 64], b1[4 * C + 32 * hb:4 * C + 32 * hb + 32])
        for wn in range(2):
            hb = 2 * c + wn
            assert np.array_equal(val[wn], w1[32 * hb:32 * hb + 32])
            assert np.array_equal(gate[wn], w1[4 * C + 32 * hb:4 * C + 32 * hb + 32])
        got = np.zeros((C, 64), np.float16)
        for i in range(2):                                   # XF_TILE(2, 5, 10, GB + (wm * 4 + 2 i) * 1024, 5 * wn, acc)
            pieces, _ = _tile(blk.tail_stream, t0 + 4 + i)
            for ksl in range(2):
                for j in range(10):
                    frag = pieces[ksl * 10 + j]
                    for lane in range(64):
                        lq, hi = lane & 31, lane >> 5
                        got[32 * j + lq, 16 * (2 * i + ksl) + 8 * hi:][:8] = frag[lane]
        assert np.array_equal(got, w2[:, 64 * c:64 * c + 64])
    prm = blk.tail_prm.numpy()
    for row, exp in enumerate((w["out1.b"].numpy(), fb("q2.w", "norm2"), w["out2.b"].numpy(), w["ff2.b"].numpy(),
                               w["proj_out.b"].numpy())):
        assert np.array_equal(prm[row], exp), row
    assert np.array_equal(blk.head_prm.numpy(), np.stack([w["proj_in.b"].numpy(), fb("q1.w", "norm1"), fb("k1.w", "norm1"),
                                                          fb("v1.w", "norm1")]))


def _xoff(rowblk, kst, col, lq):  # xformer.hip: xoff()
    return ((rowblk * kst + (col >> 4)) * 2 + ((col >> 3) & 1)) * 512 + lq * 16 + ((col >> 2) & 1) * 8


def test_operand_image_offsets_are_consistent():
    """An accumulator quad written at xoff() is read back by the A-side fragment read `(rowblk * kst + kstep) * 1024 +
    lane * 16` as columns 16 kstep + 8 hi + e of row lq."""
    for kst in (20, 4):
        seen = set()
        for rb in range(4):
            for col in range(0, 16 * kst, 4):
                for lq in range(32):
                    off = _xoff(rb, kst, col, lq)
                    kstep, hi, e0 = col // 16, (col // 8) & 1, col % 8
                    assert off == (rb * kst + kstep) * 1024 + (hi * 32 + lq) * 16 + e0 * 2
                    assert off % 8 == 0 and off not in seen
                    seen.add(off)
        assert len(seen) * 8 == 4 * 32 * 16 * kst * 2  # the image is covered exactly once


def test_context_fragments_are_what_the_kernel_reads():
    B, heads, Lk = 2, 5, 77
    g = torch.Generator().manual_seed(3)
    k = torch.randn(B, Lk, C, generator=g).half()
    vt = torch.randn(B, C, 80, generator=g).half()
    kf, vf = xformer.pack_context_frags(k, vt, Lk, heads)
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


def _small_unet():
    from diffbir_amd.model.unet import ControlledUnetModel
    cfg = configs._unet(320, 64, mult=(1,), attn=(1,), nrb=1)
    net = ControlledUnetModel(**cfg)
    net.load_state_dict(synth_state_dict(net._spec, seed=5), strict=True)
    net._dtype, net._packed = torch.float32, False
    return net


@torch.no_grad()
@pytest.mark.parametrize("pair", [None, (1, 1)])
def test_fused_block_wiring_equals_per_launch_path(monkeypatch, pair):
    """groupnorm_affine -> xf_head -> attention -> xf_tail (test double, f32) == the 16-launch path, with and without
    the shared CFG prefix (pair_bs source-row mapping, full-batch text context)."""
    emu_ops.install(monkeypatch)
    from diffbir_amd.model import unet as unet_mod
    assert unet_mod.FUSED_XF
    net = _small_unet()
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
