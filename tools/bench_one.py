#!/usr/bin/env python
"""One GEMM problem, one tile code, N launches (for rocprofv3 --pmc runs and A/B timing).
Usage: python tools/bench_one.py conv B H W Cin Cout tile [iters]   |   lin M N K tile [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DBIR_TUNING"] = "0"
from diffbir_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
DT = torch.float16


def main():
    a = sys.argv[1:]
    if a[0] == "conv":
        b, h, w, ci, co, tile = (int(x) for x in a[1:7])
        iters = int(a[7]) if len(a) > 7 else 5
        x = (torch.randn(b, h, w, ci, device=DEV)).to(DT)
        pw = ops.pack_conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DT, DEV)
        out = torch.empty(b, h, w, co, dtype=DT, device=DEV)
        fn = lambda: ops.conv3x3(x, pw, out=out, tile=tile)
        fl = 2.0 * b * h * w * co * 9 * ci
    else:
        M, N, K, tile = (int(x) for x in a[1:5])
        iters = int(a[5]) if len(a) > 5 else 5
        x = torch.randn(M, K, device=DEV).to(DT)
        pw = ops.pack_linear(torch.randn(N, K) * K ** -0.5, torch.randn(N), DT, DEV)
        out = torch.empty(M, N, dtype=DT, device=DEV)
        fn = lambda: ops.linear(x, pw, out=out, tile=tile)
        fl = 2.0 * M * N * K
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{' '.join(a)}: {us:.1f} us/launch {fl / us / 1e6:.0f} TF/s")


if __name__ == "__main__":
    main()
