#!/usr/bin/env python
"""A/B of the halo-patch convolution tiles on the UNet's stride-1 3x3 shapes (batch 16 = the CFG evaluation of 8 images):
interleaved repetitions, min of HIP-event timings per launch.  python tools/probes/halo_tile_ab.py [tiles...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["DBIR_TUNING"] = "0"
from diffbir_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
DT = torch.float16
SHAPES = [  # B, H, W, Cin, Cout, residual
    (16, 64, 64, 320, 320, True), (16, 64, 64, 640, 320, False), (16, 64, 64, 960, 320, False),
    (16, 32, 32, 640, 640, True), (16, 32, 32, 320, 640, False), (16, 32, 32, 1280, 640, False), (16, 32, 32, 960, 640, False),
    (16, 16, 16, 1280, 1280, True), (16, 16, 16, 640, 1280, False), (16, 16, 16, 2560, 1280, False),
    (8, 64, 64, 320, 320, True), (8, 32, 32, 640, 640, True),
]


def main():
    tiles = [int(t) for t in sys.argv[1:]] or [50, 52]
    print("shape".ljust(34) + "".join(f"t{t:<3d} us   TF/s   " for t in tiles))
    for b, h, w, ci, co, res in SHAPES:
        x = torch.randn(b, h, w, ci, device=DEV).to(DT)
        pw = ops.pack_conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DT, DEV)
        r = torch.randn(b, h, w, co, device=DEV).to(DT) if res else None
        emb = torch.randn(b, co, device=DEV).to(DT)
        out = torch.empty(b, h, w, co, dtype=DT, device=DEV)
        fl = 2.0 * b * h * w * co * 9 * ci
        best = {t: float("inf") for t in tiles}
        ok = {}
        for rep in range(6):
            for t in tiles:
                try:
                    for _ in range(2 if rep == 0 else 1):
                        ops.conv3x3(x, pw, residual=r, rowvec=emb, out=out, tile=t)
                except Exception:
                    ok[t] = False
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.conv3x3(x, pw, residual=r, rowvec=emb, out=out, tile=t)
                e1.record()
                e1.synchronize()
                best[t] = min(best[t], e0.elapsed_time(e1) * 1e3 / 5)
        line = f"conv B{b} {h}x{w} {ci}->{co}{' +res' if res else ''}".ljust(34)
        for t in tiles:
            line += ("   n/a          " if ok.get(t) is False else f"{best[t]:7.1f} {fl / best[t] * 1e-6:6.0f}   ")
        print(line, flush=True)


if __name__ == "__main__":
    main()
