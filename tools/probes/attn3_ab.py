#!/usr/bin/env python
"""attn3_kernel (8 waves, wave groups alternating matrix / softmax phases) against attn2_kernel on the UNet's self-attention shapes:
bit-for-bit comparison is not expected (same arithmetic, same order -> in fact expected equal; reported), interleaved min-of-6 timings.
python tools/probes/attn3_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffbir_amd import native, ops  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [(16, 5, 4096, 4096), (8, 5, 4096, 4096), (16, 10, 1024, 1024), (8, 10, 1024, 1024), (16, 20, 256, 256), (2, 5, 4096, 4096),
          (4, 5, 16384, 16384)]
for dt in (torch.float16, torch.bfloat16):
    for B, H, Lq, Lk in SHAPES:
        C = 64 * H
        q, k = torch.randn(B, Lq, C, device=DEV).to(dt), torch.randn(B, Lk, C, device=DEV).to(dt)
        vt = torch.randn(B, C, Lk, device=DEV).to(dt)
        outs, best = {}, {}
        for v in (2, 8):
            native.check(native.lib().dbir_set_option(1, v), "set_option")
            o = torch.empty(B, Lq, C, dtype=dt, device=DEV)
            ops.attention(q, k, vt, o, H, Lk, 0.125)
            torch.cuda.synchronize()
            outs[v], best[v] = o, float("inf")
        for rep in range(6):
            for v in (2, 8):
                native.check(native.lib().dbir_set_option(1, v), "set_option")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ops.attention(q, k, vt, outs[v], H, Lk, 0.125)
                e1.record()
                e1.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1) * 1e3 / 3)
        fl = 4.0 * B * H * Lq * Lk * 64
        ref = torch.softmax((q.float().reshape(B, Lq, H, 64).permute(0, 2, 1, 3) @ k.float().reshape(B, Lk, H, 64).permute(0, 2, 3, 1)) * 0.125, -1) \
            @ vt.float().reshape(B, H, 64, Lk).transpose(-1, -2) if B * H * Lq * Lk <= 16 * 5 * 4096 * 4096 // 4 else None
        err = "" if ref is None else f"  max err vs f32 {(outs[8].float().reshape(B, Lq, H, 64).permute(0, 2, 1, 3) - ref).abs().max().item():.2e}"
        print(f"{str(dt)[6:]:9s} B{B} H{H} Lq{Lq} Lk{Lk}: attn2 {best[2]:8.1f} us ({fl / best[2] * 1e-6:5.0f} TF/s)   attn3 {best[8]:8.1f} us ({fl / best[8] * 1e-6:5.0f} TF/s)"
              f"   equal {bool(torch.equal(outs[2], outs[8]))}{err}", flush=True)
native.check(native.lib().dbir_set_option(1, 2), "set_option")
