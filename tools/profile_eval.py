#!/usr/bin/env python
"""Per-shape time table of ONE batched network evaluation (ControlNet + UNet, 2B = 16 samples, 64x64 latent) on one
MI355X: every MFMA launch bracketed by HIP events on the launch stream, aggregated by problem shape.
Usage: python tools/profile_eval.py [--batch 8] [--json out.json]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffbir_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--json", default=None)
    ap.add_argument("--pair", action="store_true", help="the CFG evaluation the samplers issue: [uncond || cond] with cfg_pair")
    ap.add_argument("--pmc-mode", action="store_true", help="under rocprofv3 --pmc: 1 warm + 1 evaluation, no tables")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pipe, cldm, swin = bench.build_engine(dev, torch.float16)
    cldm.overlap_streams = False
    cldm.use_graph = False   # per-launch instrumentation (HIP events / PMC per kernel): eager launches, never a replay
    B2 = 2 * a.batch
    x = torch.randn(B2, 4, 64, 64, device=dev)
    cond = dict(c_txt=torch.randn(B2, 77, 1024, device=dev), c_img=torch.randn(B2, 4, 64, 64, device=dev))
    if a.pair:
        x = torch.randn(a.batch, 4, 64, 64, device=dev).repeat(2, 1, 1, 1)
        cond["c_img"] = torch.randn(a.batch, 4, 64, 64, device=dev).repeat(2, 1, 1, 1)
        cond["cfg_pair"] = (1, a.batch)
    t = torch.full((B2,), 500.0, device=dev)
    for _ in range(1 if a.pmc_mode else 2):
        cldm(x, t, cond)
    torch.cuda.synchronize()
    if a.pmc_mode:
        cldm(x, t, cond)
        torch.cuda.synchronize()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        cldm(x, t, cond)
    e1.record()
    torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / 3
    ops.start_profile()
    cldm(x, t, cond)
    torch.cuda.synchronize()
    rec = ops.stop_profile()
    agg = {}
    for kind, flops, s0, s1, tag, _nb in rec:
        r = agg.setdefault(tag, [0, 0.0, 0.0])
        r[0] += 1
        r[1] += s0.elapsed_time(s1) * 1e-3
        r[2] += flops
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot_t = sum(v[1] for v in agg.values())
    tot_f = sum(v[2] for v in agg.values())
    print(f"one eval (batch {B2}): wall {wall:.2f} ms un-instrumented; MFMA launches: {tot_t*1e3:.2f} ms, "
          f"{tot_f/1e12:.2f} TFLOP, {tot_f/tot_t/1e12:.0f} TF/s average")
    for tag, (n, sec, fl) in rows[:60]:
        print(f"{tag:58s} x{n:3d} {sec*1e3:8.3f} ms {sec/tot_t*100:5.1f}% {fl/sec/1e12:7.0f} TF/s  {sec/n*1e6:7.1f} us/launch")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(dict(wall_ms=wall, rows=[dict(tag=k, n=v[0], sec=v[1], flops=v[2]) for k, v in rows]), f, indent=1)


if __name__ == "__main__":
    main()
