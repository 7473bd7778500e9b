#!/usr/bin/env python
"""BASELINE.json configs[3]: tiled sampling (mixture of diffusers), 1x2048x2048 input, tile 512 / stride 256, 50-step
SpacedSampler + CFG, fp16, one MI355X.  Prints seconds per image and tile-evaluations per second (parity of this
configuration is covered at small size by tests/test_pipeline_gpu.py::*tiled*)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pipe, cldm, swin = bench.build_engine(dev, torch.float16)
    lq = torch.as_tensor(np.random.RandomState(0).randint(0, 256, (1, a.size, a.size, 3)).astype(np.uint8)).to(dev)
    args = (a.steps, 1.0, False, 512, 256, False, 256, False, 256, True, 512, 256, "", bench.NEG, 4.0, "noise", "spaced",
            0, False, 0, 0, 300, 1, 1, 1)
    torch.manual_seed(231)
    pipe.run(lq, 2, *args[1:])          # warm-up: 2 steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.run(lq, *args)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lat = a.size // 8
    ntile = len(range(0, lat - 64 + 1, 32)) ** 2
    print(json.dumps(dict(config=f"tiled {a.size}x{a.size}, tile 512/256, {a.steps} steps, CFG 4.0, fp16, 1 GPU",
                          seconds_per_image=dt, tiles_per_eval=ntile,
                          tile_sample_evals_per_s=ntile * 2 * a.steps / dt, out_shape=list(out.shape),
                          peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30)))


if __name__ == "__main__":
    main()
