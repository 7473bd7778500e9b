"""GPU check of the first-use autotune (diffbir_amd/autotune.py): an image shape / batch the shipped table does not hold
(768x640, batch 3) — pass 1 tunes every missed key on first use, pass 2 runs from the cache; then the same shape with
DBIR_AUTOTUNE disabled (nearest-M fallback) for comparison.  python tools/probes/autotune_miss_check.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["DBIR_AUTOTUNE_CACHE"] = "/tmp/dbir_autotune_check"
import bench  # noqa: E402
from diffbir_amd import autotune  # noqa: E402

dev = torch.device("cuda:0")
pipe, cldm, swin = bench.build_engine(dev, torch.float16)
lq = torch.as_tensor(np.random.RandomState(0).randint(0, 256, (3, 768, 640, 3)).astype(np.uint8)).to(dev)


def timed(steps=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_once(pipe, lq, steps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


autotune.ENABLED = False
timed(2)
t_off = min(timed(), timed())
autotune.ENABLED = True
t_first = timed()
n_tuned = autotune.stats["tuned"]
t_second = min(timed(), timed())
print(f"768x640 batch 3, 10 spaced steps: nearest-M fallback {t_off:.3f} s | first call with on-miss autotune {t_first:.3f} s "
      f"({n_tuned} keys tuned) | second call {t_second:.3f} s  ({(t_off / t_second - 1) * 100:+.1f} % vs fallback); "
      f"cache hits {autotune.stats['hits']}")
