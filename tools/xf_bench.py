"""Timing of the fused transformer kernels alone (GPU): xf_tail / xf_head at the benchmark's shapes, min of N HIP-event
timings, for every staging variant given.  python tools/xf_bench.py [variants...]"""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_kernels_gpu import _xf_weights
from diffbir_amd import native, ops

DEV = torch.device("cuda:0")
dtype = torch.float16
C, L, Lk = 320, 4096, 77
variants = [int(v) for v in sys.argv[1:]] or [0, 1]
blk = ops.pack_xf_block(_xf_weights(), dtype, DEV)


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


for B in (16, 8):
    M = B * L
    attn, h = torch.randn(M, C, device=DEV).to(dtype), torch.randn(M, C, device=DEV).to(dtype)
    x = torch.randn(B, 64, 64, C, device=DEV).to(dtype)
    k, vt = torch.randn(B, Lk, C, device=DEV).to(dtype), torch.randn(B, C, 80, device=DEV).to(dtype)
    kf, vf = ops.pack_context_frags(k, vt, Lk, 5)
    ab = ops.groupnorm_affine(x, torch.ones(C, device=DEV), torch.zeros(C, device=DEV), 1e-6)
    out = torch.empty_like(x)
    for v in variants:
        native.check(native.lib().dbir_set_option(2, v), "set_option")
        t_tail = timeit(lambda: ops.xf_tail(attn, h, x, blk, kf, vf, Lk, 0.125, L, out=out))
        t_head = timeit(lambda: ops.xf_head(x, ab, blk, L))
        fl_t, fl_h = 2.0 * M * C * 16 * C, 2.0 * M * C * 4 * C
        print(f"B{B} variant {v}: xf_tail {t_tail:7.1f} us ({fl_t / t_tail / 1e6:6.0f} TF/s)   xf_head {t_head:6.1f} us "
              f"({fl_h / t_head / 1e6:5.0f} TF/s)")
