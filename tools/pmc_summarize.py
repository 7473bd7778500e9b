#!/usr/bin/env python
"""Summarise the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into per-kernel HBM bytes per launch.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B for wide coalesced
streams, MI355X_MICROARCH.md §HBM).  Only dispatches between the two marker kernels... (none: the profiled command runs
exactly one network evaluation after warm-up with counters, so all dispatches of a GEMM kernel name are averaged)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(outdir, counter):
    rows = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(outdir, counter, "**", "*counter_collection.csv"), recursive=True)
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                name = r["Kernel_Name"]
                rows[name][0] += 1
                rows[name][1] += float(r["Counter_Value"])
    return rows, files


def family(name):
    if "gemm_" in name or "gemm8p_" in name or "splitk_reduce" in name or "xf_head_kernel" in name or "xf_tail_kernel" in name or "xf2_head_kernel" in name or "xf2_tail_kernel" in name:
        return "gemm"   # the implicit-GEMM family incl. the fused transformer kernels (bench.py counts them in it)
    if "attn" in name:
        return "attention"
    if "gn_" in name or "ln_kernel" in name:
        return "norm"
    return None


def main():
    outdir, dst = sys.argv[1], sys.argv[2]
    fetch, f1 = read(outdir, "FETCH_SIZE")
    write, f2 = read(outdir, "WRITE_SIZE")
    fam = defaultdict(lambda: dict(launches=0, fetch_kib=0.0, write_kib=0.0))
    per = {}
    for name in set(fetch) | set(write):
        fa = family(name)
        n = fetch.get(name, [0, 0])[0] or write.get(name, [0, 0])[0]
        fk, wk = fetch.get(name, [0, 0.0])[1], write.get(name, [0, 0.0])[1]
        if fa:
            fam[fa]["launches"] += n
            fam[fa]["fetch_kib"] += fk
            fam[fa]["write_kib"] += wk
            per[name[:120]] = dict(launches=n, bytes_per_launch=(2.0 * fk + wk) * 1024.0 / max(n, 1))
    res = dict(method="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace); bytes = "
                      "(2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count correction)",
               workload="tools/profile_eval.py --pmc-mode: warm-up + timed evaluations of ControlNet+UNet, 16 samples",
               files=[os.path.relpath(x, outdir) for x in f1 + f2], families={}, kernels=per)
    for fa, v in fam.items():
        n = max(v["launches"], 1)
        res["families"][fa] = dict(launches=v["launches"], fetch_bytes_per_launch_corrected=2.0 * v["fetch_kib"] * 1024 / n,
                                   write_bytes_per_launch=v["write_kib"] * 1024 / n,
                                   bytes_per_launch=(2.0 * v["fetch_kib"] + v["write_kib"]) * 1024 / n)
    g = res["families"].get("gemm", {})
    res["evaluations_profiled"] = 2   # tools/profile_eval.py --pmc-mode: one warm-up + one evaluation
    res["gemm_bytes_per_kernel_launch"] = g.get("bytes_per_launch")
    # per LOGICAL GEMM launch of bench.py's roofline record (a split-K GEMM = main kernel + reduce kernel):
    res["gemm_bytes_per_eval"] = g.get("bytes_per_launch", 0.0) * g.get("launches", 0) / 2.0
    res["gemm_bytes_per_launch"] = None  # filled by bench.py: gemm_bytes_per_eval / logical launches per evaluation
    with open(dst, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["families"], indent=1))


if __name__ == "__main__":
    main()
