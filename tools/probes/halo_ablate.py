#!/usr/bin/env python
"""Ablation timing of the halo-patch conv kernel (tile 50) on the MI355X: full kernel vs. no epilogue (60) / no MFMAs
(61) / no steady-state staging (62) / no fragment reads (63), against the de-phased generic kernel (37), for several
K depths and grid sizes; fits  t(block) = a + b * K_tiles  to separate prologue + epilogue from the K loop.
Usage: python tools/probes/halo_ablate.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["DBIR_TUNING"] = "0"
from diffbir_amd import ops  # noqa: E402

DEV, DT = torch.device("cuda:0"), torch.float16


def time_us(fn, iters=8):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def main():
    print(f"{'shape':34s} " + " ".join(f"{n:>8s}" for n in ("t37", "t50", "noEpi", "noMFMA", "noStage", "noFrag", "t52(LS)", "noEpi", "noMFMA",
                                                          "noStage", "noFrag", "t52+res")))
    for (b, hw, ci, co) in [(16, 64, 64, 320), (16, 64, 128, 320), (16, 64, 320, 320), (16, 64, 640, 320),
                            (16, 64, 960, 320), (8, 64, 320, 320), (8, 64, 640, 320), (16, 32, 640, 640),
                            (16, 32, 1280, 640), (16, 16, 1280, 1280)]:
        x = torch.randn(b, hw, hw, ci, device=DEV).to(DT)
        pw = ops.pack_conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DT, DEV)
        out = torch.empty(b, hw, hw, co, dtype=DT, device=DEV)
        res = torch.randn(b, hw, hw, co, device=DEV).to(DT)
        emb = torch.randn(b, co, device=DEV).to(DT)
        row = []
        for tile in (37, 50, 60, 61, 62, 63, 52, 64, 65, 66, 67):
            row.append(time_us(lambda: ops.conv3x3(x, pw, out=out, tile=tile)))
        row.append(time_us(lambda: ops.conv3x3(x, pw, out=out, residual=res, rowvec=emb, tile=52)))
        fl = 2.0 * b * hw * hw * co * 9 * ci
        tiles = (b * hw * hw // 256) * (co // 160)
        print(f"conv {b}x{hw}x{hw} {ci}->{co} ({tiles:4d} blk, {ci // 64 * 9:3d} kt) " + " ".join(f"{v:8.1f}" for v in row) +
              f"   t50 {fl / row[1] / 1e6:.0f} / t52 {fl / row[6] / 1e6:.0f} TF/s")


if __name__ == "__main__":
    main()
