#!/usr/bin/env python
"""Where one C2 pass (batch 8 x 512 x 512, 50 spaced steps, CFG) spends its wall time OUTSIDE the 50 network evaluations:
synchronised wall-clock timers around the pipeline's stages (cleaner, condition = VAE encode + CLIP, the sampler loop, VAE
decode, colour fix + device-to-host copy).  The synchronisations cost a few hundred microseconds per stage; the un-instrumented
pass is timed first for comparison.   python tools/stage_times.py [--batch 8] [--steps 50]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pipe, cldm, swin = bench.build_engine(dev, torch.float16)[:3]
    import numpy as np
    lq = np.random.RandomState(100).randint(0, 256, (a.batch, 512, 512, 3)).astype(np.uint8)

    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_once(pipe, lq, a.steps)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    once()
    plain = [once() for _ in range(2)]
    print(f"un-instrumented pass: {min(plain):.1f} ms (runs: {', '.join(f'{v:.1f}' for v in plain)})")

    acc = {}

    def timed(name, fn):
        def wrapper(*args, **kw):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*args, **kw)
            torch.cuda.synchronize()
            acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
            return out
        return wrapper

    from diffbir_amd import pipeline as pl
    from diffbir_amd.sampler import spaced_sampler
    pipe.apply_cleaner = timed("cleaner (SwinIR stage 1, incl. padding)", pipe.apply_cleaner)
    cldm.prepare_condition = timed("condition: VAE encode + CLIP(pos)", cldm.prepare_condition)
    cldm.clip.encode = timed("  CLIP text encoder (pos + neg)", cldm.clip.encode)
    cldm.vae_decode = timed("VAE decode", cldm.vae_decode)
    orig_sample = spaced_sampler.SpacedSampler.sample
    spaced_sampler.SpacedSampler.sample = timed("sampler loop (50 evaluations + updates)", orig_sample)
    orig_wave = pl.wavelet_reconstruction
    pl.wavelet_reconstruction = timed("wavelet colour fix", orig_wave)
    try:
        total = once()
    finally:
        spaced_sampler.SpacedSampler.sample = orig_sample
        pl.wavelet_reconstruction = orig_wave
    print(f"instrumented pass: {total:.1f} ms")
    named = 0.0
    for k, v in acc.items():
        print(f"  {k:48s} {v:8.1f} ms  {100 * v / total:5.1f} %")
        if not k.startswith("  "):
            named += v
    print(f"  {'rest (uint8 -> tensor, output conversion + D2H, host)':48s} {total - named:8.1f} ms  {100 * (total - named) / total:5.1f} %")


if __name__ == "__main__":
    main()
