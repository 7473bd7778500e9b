#!/usr/bin/env python
"""Register-streaming linear kernel (csrc/gemm_rs.hip, tiles 93 - 97) against the shipped table's choice on the linears of the
16x16 / 8x8 / 32x32 / 64x64 latent levels at batch 16: correctness against an f32 matmul, then interleaved min-of-6 HIP-event
timings.  python tools/probes/rs_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DBIR_AUTOTUNE", "0")
from diffbir_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
DT = torch.float16
LIN = [  # M, N, K, residual
    (4096, 1280, 1280, True), (4096, 1280, 1280, False), (4096, 2560, 1280, False), (4096, 1280, 5120, True),
    (4096, 1280, 2560, False), (4096, 1280, 640, False), (4096, 1280, 1920, False),
    (1024, 1280, 1280, True), (1024, 1280, 1280, False), (1024, 2560, 1280, False), (1024, 1280, 5120, True), (1024, 1280, 2560, False),
    (16384, 640, 640, True), (16384, 640, 1920, False), (16384, 640, 320, False), (16384, 1280, 640, False),
    (65536, 320, 320, True), (65536, 320, 640, False), (65536, 320, 960, False),
    (2048, 1280, 1280, True), (8192, 640, 640, True),
]
TILES = [0, 93, 94, 95, 96, 97]


def main():
    print("shape".ljust(34) + "".join(f"t{t:<3d} us   TF/s   " for t in TILES))
    for M, N, K, res in LIN:
        x = torch.randn(M, K, device=DEV).to(DT)
        w, b = torch.randn(N, K) * K ** -0.5, torch.randn(N)
        pw = ops.pack_linear(w, b, DT, DEV)
        r = torch.randn(M, N, device=DEV).to(DT) if res else None
        ref = x.float() @ w.to(DEV).to(DT).float().t() + b.to(DEV)
        if res:
            ref = ref + r.float()
        out = torch.empty(M, N, dtype=DT, device=DEV)
        best, ok = {}, {}
        for t in TILES:
            try:
                out.zero_()
                ops.linear(x, pw, residual=r, out=out, tile=t)
                torch.cuda.synchronize()
                err = (out.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
                ok[t] = err < 4e-3
                if not ok[t]:
                    print(f"   tile {t}: MISMATCH rel err {err:.3g}")
            except Exception as e:  # noqa: BLE001
                ok[t] = None
            best[t] = float("inf")
        for rep in range(6):
            for t in TILES:
                if not ok[t]:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.linear(x, pw, residual=r, out=out, tile=t)
                e1.record()
                e1.synchronize()
                best[t] = min(best[t], e0.elapsed_time(e1) * 1e3 / 5)
        line = f"lin M{M} N{N} K{K}{' +res' if res else ''}".ljust(34)
        fl = 2.0 * M * N * K
        for t in TILES:
            line += ("   n/a          " if ok[t] is None else ("   BAD          " if not ok[t] else f"{best[t]:7.1f} {fl / best[t] * 1e-6:6.0f}   "))
        print(line, flush=True)


if __name__ == "__main__":
    main()
