#!/usr/bin/env python
"""One GEMM problem, one tile code, N launches (for rocprofv3 --pmc runs and A/B timing).
Usage: python tools/probes/bench_one.py conv B H W Cin Cout tile [iters]   |   lin M N K tile [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    if os.environ.get("DBIR_GEMM_DEBUG") == "5":
        import ctypes
        ws = torch.zeros(1 << 22, dtype=torch.int64, device=DEV)
        orig = ops.apply_tile_code

        def patched(d, code, device):
            orig(d, code, device)
            d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 8
        ops.apply_tile_code = patched
        fn()
        torch.cuda.synchronize()
        v = ws.reshape(-1, 8)
        v = v[v[:, 5] > 0].double()
        nk = v[:, 5].mean().item()
        if tile in (36, 37, 38):
            for g in (0, 1):
                vg = v[v[:, 6] == g]
                print(f" group {g}: stage(+vmcnt) {vg[:,0].mean().item()/nk:.0f}  barrier1 {vg[:,1].mean().item()/nk:.0f}  "
                      f"reads+MFMA(+vmcnt0 for g1) {vg[:,2].mean().item()/nk:.0f}  barrier2 {vg[:,3].mean().item()/nk:.0f}  "
                      f"total {vg[:,4].mean().item()/nk:.0f} cycles per K tile")
            return
        names = (["load part (reads+glds+vmcnt)", "barrier1+lgkmcnt", "MFMA issue", "barrier2", "loop total"] if tile == 13 else
                 ["vmcnt wait", "barrier wait", "stage issue", "reads+MFMA issue", "loop total"])
        print(f"instrumented waves: {v.shape[0]}, K tiles per block {nk:.0f}; cycles per K tile (s_memtime ticks):")
        for i, n in enumerate(names):
            print(f"   {n:30s} mean {v[:, i].mean().item() / nk:8.0f}   min {v[:, i].min().item() / nk:8.0f}   max {v[:, i].max().item() / nk:8.0f}")
        return
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
