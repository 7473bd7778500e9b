"""Probe: reference counting of torch.cuda.MemPool across use_mem_pool contexts and at destruction (round 5: a plan's private
pool aborted the process when it was garbage-collected)."""
import gc
import sys
import torch

def trial(n_ctx, keep_tensor):
    pool = torch.cuda.MemPool()
    print("  created: use_count", pool.use_count(), flush=True)
    keep = []
    for i in range(n_ctx):
        with torch.cuda.use_mem_pool(pool):
            t = torch.empty(1 << 20, device="cuda")
            if keep_tensor:
                keep.append(t)
        print(f"  after context {i}: use_count", pool.use_count(), flush=True)
    del pool
    gc.collect()
    print("  pool deleted", "(tensor from it still alive)" if keep else "", flush=True)
    return keep

which = sys.argv[1]
print("trial", which, flush=True)
if which == "a":
    trial(1, False)
elif which == "b":
    trial(2, False)
elif which == "c":
    k = trial(1, True)
    print("  tensor sum", float(k[0].zero_().sum()))
elif which == "d":
    k = trial(2, True)
    print("  tensor sum", float(k[0].zero_().sum()))
print("OK", which, flush=True)
