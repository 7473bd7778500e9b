"""Probe: create / drop many recorded plans (private torch.cuda.MemPool each) with garbage collection in between — the full GPU
suite aborted inside a GC pass while a new plan was being built (round 5, call 11)."""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import build_engine  # noqa: E402
from diffbir_amd.model import cldm as cm  # noqa: E402

dev = torch.device("cuda:0")
B = 2
for rep in range(6):
    pipe, cldm, swin = build_engine("tiny", "DIFFUSION_V21", dev, torch.float16)
    D = cldm.unet.cfg["context_dim"]
    for k in range(4):
        x = torch.randn(2 * B, 4, 64, 64, device=dev)
        ci = torch.randn(2 * B, 4, 64, 64, device=dev)
        t = torch.full((2 * B,), 100.0 + k, device=dev)
        c_txt = torch.randn(2 * B, 77, D, device=dev)
        ev = cm._EvalPlan(cldm, x, t, c_txt, ci, None)
        out = ev.run(x, t, ci)
        torch.cuda.synchronize()
        del ev
        gc.collect()
    del pipe, cldm, swin
    gc.collect()
    torch.cuda.empty_cache()
    print("rep", rep, "ok", flush=True)
print("OK")
