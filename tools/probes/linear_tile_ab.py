#!/usr/bin/env python
"""A/B of GEMM tile variants on the UNet's transformer linears and the non-halo convolutions at batch 16 (the CFG evaluation
of 8 images): the shipped table's choice (tile 0 -> tuning_gfx950.json) against the tiles given (default: the producer /
consumer tiles 90 - 92).  Interleaved repetitions, min of HIP-event timings.  python tools/probes/linear_tile_ab.py [tiles...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DBIR_AUTOTUNE", "0")
from diffbir_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
DT = torch.float16
LIN = [  # M, N, K, residual, geglu
    (4096, 1280, 1280, True, False), (4096, 1280, 1280, False, False), (4096, 2560, 1280, False, False),
    (4096, 10240, 1280, False, True), (4096, 1280, 5120, True, False),
    (16384, 640, 640, True, False), (16384, 1280, 640, False, False), (16384, 5120, 640, False, True),
    (16384, 640, 2560, True, False), (65536, 320, 320, True, False), (65536, 320, 640, False, False),
    (1024, 1280, 1280, True, False), (1024, 10240, 1280, False, True), (1024, 1280, 5120, True, False),
    (2048, 1280, 1280, True, False), (8192, 640, 640, True, False),
]
CONV = [  # B, H, W, Cin, Cout, stride, upsample
    (16, 64, 64, 320, 320, 2, False), (16, 32, 32, 640, 640, 2, False), (16, 16, 16, 1280, 1280, 2, False),
    (16, 32, 32, 640, 640, 1, True), (16, 16, 16, 1280, 1280, 1, True), (16, 8, 8, 1280, 1280, 1, True),
    (16, 8, 8, 1280, 1280, 1, False), (16, 8, 8, 2560, 1280, 1, False),
]


def timeit(fn, tiles):
    best, ok = {t: float("inf") for t in tiles}, {}
    for rep in range(6):
        for t in tiles:
            if ok.get(t) is False:
                continue
            try:
                for _ in range(2 if rep == 0 else 1):
                    fn(t)
            except Exception:
                ok[t] = False
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn(t)
            e1.record()
            e1.synchronize()
            best[t] = min(best[t], e0.elapsed_time(e1) * 1e3 / 5)
    return best, ok


def main():
    tiles = [0] + ([int(t) for t in sys.argv[1:]] or [90, 91, 92])
    print("shape".ljust(44) + "".join(f"t{t:<3d} us   TF/s   " for t in tiles))
    for M, N, K, res, geglu in LIN:
        x = torch.randn(M, K, device=DEV).to(DT)
        if geglu:
            pw = ops.pack_geglu(torch.randn(N, K) * K ** -0.5, torch.randn(N), DT, DEV)
            out = torch.empty(M, N // 2, dtype=DT, device=DEV)
        else:
            pw = ops.pack_linear(torch.randn(N, K) * K ** -0.5, torch.randn(N), DT, DEV)
            out = torch.empty(M, N, dtype=DT, device=DEV)
        r = torch.randn(M, N, device=DEV).to(DT) if res else None
        best, ok = timeit(lambda t: ops.linear(x, pw, residual=r, out=out, tile=t), tiles)
        line = f"lin M{M} N{N} K{K}{' +res' if res else ''}{' geglu' if geglu else ''}".ljust(44)
        fl = 2.0 * M * N * K
        for t in tiles:
            line += ("   n/a          " if ok.get(t) is False else f"{best[t]:7.1f} {fl / best[t] * 1e-6:6.0f}   ")
        print(line, flush=True)
    for b, h, w, ci, co, st, up in CONV:
        x = torch.randn(b, h, w, ci, device=DEV).to(DT)
        pw = ops.pack_conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DT, DEV)
        ho, wo = (2 * h, 2 * w) if up else (h // st, w // st)
        out = torch.empty(b, ho, wo, co, dtype=DT, device=DEV)
        best, ok = timeit(lambda t: ops.conv3x3(x, pw, stride=st, pad=1, upsample=up, out=out, tile=t), tiles)
        fl = 2.0 * b * ho * wo * co * 9 * ci
        line = f"conv B{b} {h}x{w} {ci}->{co}{' s2' if st == 2 else ''}{' up' if up else ''}".ljust(44)
        for t in tiles:
            line += ("   n/a          " if ok.get(t) is False else f"{best[t]:7.1f} {fl / best[t] * 1e-6:6.0f}   ")
        print(line, flush=True)


if __name__ == "__main__":
    main()
