#!/usr/bin/env python
"""A/B of the fine-phase 256x320 kernel (tile 80, gemm_8p.hip) against the tuned incumbents on the UNet's GEMM shapes of a
batch-16 evaluation (the CFG evaluation of 8 images): interleaved repetitions in one process, min and median of HIP-event
timings per launch, random operands.   python tools/probes/p8_ab.py [--lin] [--quick]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["DBIR_AUTOTUNE"] = "0"
from diffbir_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
DT = torch.float16
# B, H, W, Cin, Cout, residual, upsample, candidate codes (0 = the shipped tuning table's choice)
CONVS = [
    (16, 64, 64, 320, 320, True, False, [0, 50, 80]),
    (16, 64, 64, 640, 320, False, False, [0, 50, 80]),
    (16, 64, 64, 960, 320, False, False, [0, 80]),
    (16, 32, 32, 640, 640, True, False, [0, 50, 80, 280]),
    (16, 32, 32, 320, 640, False, False, [0, 80, 280]),
    (16, 32, 32, 1280, 640, False, False, [0, 80, 280]),
    (16, 32, 32, 1920, 640, False, False, [0, 80, 280]),
    (16, 16, 16, 1280, 1280, True, False, [0, 250, 280, 480]),
    (16, 16, 16, 640, 1280, False, False, [0, 280, 480]),
    (16, 16, 16, 2560, 1280, False, False, [0, 250, 280, 480]),
    (16, 8, 8, 1280, 1280, True, False, [0, 480, 880]),
    (16, 32, 32, 640, 640, False, True, [0, 80]),       # decoder upsample conv 32 -> 64 (M = 65536)
    (16, 16, 16, 1280, 1280, False, True, [0, 80, 280]),  # 16 -> 32 (M = 16384)
    (8, 64, 64, 320, 320, True, False, [0, 50, 80, 280]),
]
# M, N, K, residual, codes
LINS = [
    (65536, 320, 320, True, [0, 80]), (65536, 320, 1280, True, [0, 80]), (16384, 640, 640, True, [0, 80, 280]),
    (16384, 640, 2560, True, [0, 80, 280]), (4096, 1280, 1280, True, [0, 80, 280, 480]), (4096, 1280, 5120, True, [0, 280, 480]),
    (4096, 2560, 1280, False, [0, 80, 280]), (16384, 1280, 640, False, [0, 80]), (65536, 320, 640, False, [0, 80]),
    # the feed-forward GEMMs of the 32x32 / 16x16 transformer blocks as PLAIN linears (same MFMA work as the GEGLU form)
    (16384, 5120, 640, False, [0, 10, 30, 80]), (4096, 10240, 1280, False, [0, 10, 30, 80]),
    (16384, 640, 2560, True, [0, 37, 50, 80, 280]), (65536, 2560, 320, False, [0, 10, 30, 80]),
]


def time_codes(fn, codes, flops, reps=7, inner=4):
    ts = {c: [] for c in codes}
    okc = []
    for c in codes:
        try:
            fn(c)
            fn(c)
            okc.append(c)
        except Exception as e:  # noqa: BLE001
            print(f"   code {c}: {str(e)[:100]}")
    torch.cuda.synchronize()
    for _ in range(reps):
        for c in okc:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                fn(c)
            e1.record()
            e1.synchronize()
            ts[c].append(e0.elapsed_time(e1) * 1e3 / inner)
    out = ""
    for c in codes:
        if c in okc:
            mn, md = min(ts[c]), statistics.median(ts[c])
            out += f" | c{c:<3d} {mn:7.1f} us {flops / mn * 1e-6:5.0f} TF (med {md:6.1f})"
        else:
            out += f" | c{c:<3d}   n/a"
    return out


def main():
    quick = "--quick" in sys.argv
    if "--lin" not in sys.argv:
        for b, h, w, ci, co, res, ups, codes in (CONVS[:4] if quick else CONVS):
            x = torch.randn(b, h, w, ci, device=DEV).to(DT)
            pw = ops.pack_conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DT, DEV)
            ho, wo = (2 * h, 2 * w) if ups else (h, w)
            r = torch.randn(b, ho, wo, co, device=DEV).to(DT) if res else None
            emb = torch.randn(b, co, device=DEV).to(DT)
            out = torch.empty(b, ho, wo, co, dtype=DT, device=DEV)
            fl = 2.0 * b * ho * wo * co * 9 * ci
            line = time_codes(lambda c: ops.conv3x3(x, pw, residual=r, rowvec=emb, out=out, tile=c, upsample=ups), codes, fl)
            print(f"conv B{b} {h}x{w} {ci}->{co}{' +res' if res else ''}{' up' if ups else ''}".ljust(36) + line, flush=True)
    if "--lin" in sys.argv or "--all" in sys.argv:
        for m, n, k, res, codes in LINS:
            x = torch.randn(m, k, device=DEV).to(DT)
            pw = ops.pack_linear(torch.randn(n, k) * k ** -0.5, torch.randn(n), DT, DEV)
            r = torch.randn(m, n, device=DEV).to(DT) if res else None
            out = torch.empty(m, n, dtype=DT, device=DEV)
            fl = 2.0 * m * n * k
            line = time_codes(lambda c: ops.linear(x, pw, residual=r, out=out, tile=c), codes, fl)
            print(f"lin M{m} N{n} K{k}{' +res' if res else ''}".ljust(36) + line, flush=True)


if __name__ == "__main__":
    main()
