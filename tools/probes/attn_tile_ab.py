#!/usr/bin/env python
"""Self-attention kernel A/B (DBIR_OPT_ATTN_VARIANT): 2 = default (FOLD softmax), 6 = the pre-round-4 softmax, 4 / 5 =
its register-budget variants.  Interleaved, min of HIP-event timings.  python tools/probes/attn_tile_ab.py [variants...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffbir_amd import native, ops  # noqa: E402

DEV = torch.device("cuda:0")
variants = [int(v) for v in sys.argv[1:]] or [2, 4, 5]
for B, H, L in ((16, 5, 4096), (8, 5, 4096), (16, 10, 1024), (8, 10, 1024), (16, 20, 256)):
    C = H * 64
    q, k = torch.randn(B, L, C, device=DEV).half(), torch.randn(B, L, C, device=DEV).half()
    vt = torch.randn(B, C, L, device=DEV).half()
    o = torch.empty_like(q)
    best = {v: 1e9 for v in variants}
    outs = {}
    for rep in range(6):
        for v in variants:
            native.check(native.lib().dbir_set_option(1, v), "set_option")
            ops.attention(q, k, vt, o, H, L, 0.125)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.attention(q, k, vt, o, H, L, 0.125)
            e1.record()
            e1.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) * 1e3 / 3)
            outs[v] = o.clone()
    fl = 4.0 * B * H * L * L * 64
    ref = outs[variants[0]].float()
    print(f"B{B} H{H} L{L}: " + "   ".join(f"variant {v}: {best[v]:7.1f} us {fl / best[v] * 1e-6:5.0f} TF/s (max diff vs first "
                                          f"{(outs[v].float() - ref).abs().max().item():.1e})" for v in variants), flush=True)
native.check(native.lib().dbir_set_option(1, 2), "set_option")
