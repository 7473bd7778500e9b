#!/usr/bin/env python
"""A/B of the persistent linear kernel's start-phase offset (DBIR_PERS_PHASE = 0 / 1 / 2, read once per process):
times the transformer-block linears of one evaluation (batch 16) on the two-workgroups-per-CU tiles 71 / 73 and on the
one-per-CU tiles 70 / 72.  Usage: DBIR_PERS_PHASE=p python tools/pers_phase_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DBIR_TUNING"] = "0"
from diffbir_amd import ops  # noqa: E402

DEV, DT = torch.device("cuda:0"), torch.float16


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ph = os.environ.get("DBIR_PERS_PHASE", "0")
    cases = [("geglu", 65536, 2560, 320), ("geglu", 16384, 5120, 640), ("geglu", 4096, 10240, 1280),
             ("lin_r", 65536, 320, 320), ("lin", 65536, 640, 320), ("lin_r", 65536, 320, 1280), ("lin_r", 16384, 640, 640),
             ("lin_r", 16384, 640, 2560), ("lin_r", 4096, 1280, 1280), ("lin_r", 4096, 1280, 5120)]
    for kind, M, N, K in cases:
        x = torch.randn(M, K, device=DEV).to(DT)
        w, b = torch.randn(N, K) * K ** -0.5, torch.randn(N) * 0.1
        pw = ops.pack_geglu(w, b, DT, DEV) if kind == "geglu" else ops.pack_linear(w, b, DT, DEV)
        n_out = N // 2 if kind == "geglu" else N
        out = torch.empty(M, n_out, dtype=DT, device=DEV)
        res = torch.randn(M, n_out, device=DEV).to(DT) if kind == "lin_r" else None
        ref = None
        row = []
        for tile in (0, 70, 71, 72, 73):
            try:
                fn = lambda: ops.linear(x, pw, out=out, residual=res, tile=tile)
                fn()
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                else:
                    assert torch.equal(out, ref) or (out.float() - ref.float()).abs().max() < 2e-2 * ref.float().abs().max(), tile
                us = timeit(fn)
                row.append(f"{tile}:{us:7.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF")
            except Exception as e:  # tile not eligible for this shape
                row.append(f"{tile}:   n/a ({type(e).__name__})")
        print(f"phase {ph} {kind:6s} {M:6d}x{N:5d}x{K:4d}  " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
