"""Timing of the fused transformer kernels alone (GPU): xf_tail / xf_head at the benchmark's shapes (C = 320: 64x64 level,
C = 640: 32x32 level), min of N HIP-event timings, next to the per-launch chain they replace (the same block through
ops.linear / layernorm / attention as model/unet.py runs it with the fused path off).  python tools/xf_bench.py [C ...]"""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_kernels_gpu import _xf_weights
from diffbir_amd import ops

DEV = torch.device("cuda:0")
dtype = torch.float16
Lk = 77
widths = [int(v) for v in sys.argv[1:]] or [320, 640]


def timeit(fn, n=8):
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3


def per_launch_tail(w, C, heads):
    """the tail as separate launches: out1 + res, LN2, q2, cross-attention, out2 + res, LN3, GEGLU, ff2 + res, proj_out + res"""
    pl = lambda n: ops.pack_linear(w[n + ".w"], w.get(n + ".b"), dtype, DEV)  # noqa: E731
    out1, q2, out2, ff2, po = pl("out1"), pl("q2"), pl("out2"), pl("ff2"), pl("proj_out")
    ff1 = ops.pack_geglu(w["ff1.w"], w["ff1.b"], dtype, DEV)
    ln = {n: (w[n + ".w"].to(DEV), w[n + ".b"].to(DEV)) for n in ("norm2", "norm3")}

    def run(attn, h, x, k, vt, L):
        B = attn.shape[0] // L
        h1 = ops.linear(attn, out1, residual=h)
        n = ops.layernorm(h1, *ln["norm2"])
        q = ops.linear(n, q2).reshape(B, L, C)
        o = torch.empty((B, L, C), dtype=dtype, device=DEV)
        ops.attention(q, k, vt, o, heads, Lk, 0.125)
        h2 = ops.linear(o.reshape(B * L, C), out2, residual=h1)
        n = ops.layernorm(h2, *ln["norm3"])
        g = ops.linear(n, ff1)
        h3 = ops.linear(g, ff2, residual=h2)
        return ops.linear(h3, po, residual=x.reshape(B * L, C))
    return run


for C in widths:
    heads = C // 64
    L = 4096 if C == 320 else 1024
    w = _xf_weights(C=C)
    blk = ops.pack_xf_block(w, dtype, DEV)
    plain = per_launch_tail(w, C, heads)
    for B in (16, 8):
        M = B * L
        attn, h = torch.randn(M, C, device=DEV).to(dtype), torch.randn(M, C, device=DEV).to(dtype)
        side = 64 if C == 320 else 32
        x = torch.randn(B, side, side, C, device=DEV).to(dtype)
        k, vt = torch.randn(B, Lk, C, device=DEV).to(dtype), torch.randn(B, C, 80, device=DEV).to(dtype)
        kf, vf = ops.pack_context_frags(k, vt, Lk, heads)
        ab = ops.groupnorm_affine(x, torch.ones(C, device=DEV), torch.zeros(C, device=DEV), 1e-6)
        out = torch.empty_like(x)
        for _ in range(2):
            plain(attn, h, x, k, vt, L)       # (first use tunes / looks up the tiles)
        t_plain = timeit(lambda: plain(attn, h, x, k, vt, L))
        fl_t, fl_h = 2.0 * M * C * 16 * C, 2.0 * M * C * 4 * C
        t_tail = timeit(lambda: ops.xf_tail(attn, h, x, blk, kf, vf, Lk, 0.125, L, out=out))
        t_head = timeit(lambda: ops.xf_head(x, ab, blk, L))
        print(f"C{C} B{B} (M {M}): xf_tail {t_tail:7.1f} us ({fl_t / t_tail / 1e6:6.0f} TF/s)   per-launch tail {t_plain:7.1f} us   "
              f"xf_head {t_head:6.1f} us ({fl_h / t_head / 1e6:5.0f} TF/s)", flush=True)
