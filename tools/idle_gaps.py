#!/usr/bin/env python
"""GPU idle-gap figure from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 1 --warmup 1 ...
    python tools/idle_gaps.py gpurun_out/trace [--window-kernel spaced_step_kernel] > profiles/r2_idle_gaps.json

Reports, over the window that spans the sampler loop (first .. last launch of `--window-kernel`, default: the whole
trace): wall span, union of kernel intervals (GPU busy with >= 1 kernel), idle = span - union, time with >= 2 kernels
resident (two-stream overlap), sum of kernel durations, number of launches, and the idle time attributed to the kernel
that FOLLOWS each gap (host could not keep up / dependency bubble), top 12.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(path):
    files = [path] if os.path.isfile(path) else sorted(
        glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                try:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                                 r.get("Queue_Id", ""), r.get("Stream_Id", "")))
                except (KeyError, ValueError):
                    continue
    rows.sort()
    return rows


def short(name):
    n = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    n = n.replace("void (anonymous namespace)::", "").replace("void ", "")
    return n[:70]


def main():
    path = sys.argv[1]
    wk = None
    if "--window-kernel" in sys.argv:
        wk = sys.argv[sys.argv.index("--window-kernel") + 1]
    rows = load(path)
    if not rows:
        print(json.dumps({"error": "no kernel trace rows found", "path": path}))
        return
    if wk:
        idx = [i for i, r in enumerate(rows) if wk in r[2]]
        if idx:
            rows = rows[idx[0]: idx[-1] + 1]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # sweep: coverage depth over time
    ev = []
    for s, e, *_ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, busy1, busy2 = 0, t0, 0, 0
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        last = t
        depth += d
    # gaps attributed to the next kernel (union-level gaps)
    gaps = defaultdict(lambda: [0, 0])
    cur_end = rows[0][1]
    hist = defaultdict(int)
    for s, e, name, *_ in rows[1:]:
        if s > cur_end:
            g = s - cur_end
            a = gaps[short(name)]
            a[0] += g
            a[1] += 1
            b = 1
            while b < g / 1000.0:
                b *= 2
            hist[f"<={b}us"] += 1
        cur_end = max(cur_end, e)
    span = t1 - t0
    streams = defaultdict(int)
    for s, e, name, q, st in rows:
        streams[f"q{q}/s{st}"] += e - s
    out = dict(
        launches=len(rows), span_ms=span / 1e6, busy_ms=busy1 / 1e6, idle_ms=(span - busy1) / 1e6,
        idle_frac=(span - busy1) / span, overlap2_ms=busy2 / 1e6, kernel_sum_ms=sum(e - s for s, e, *_ in rows) / 1e6,
        per_queue_kernel_ms={k: v / 1e6 for k, v in sorted(streams.items())},
        gap_histogram=dict(sorted(hist.items(), key=lambda kv: float(kv[0][2:-2]))),
        idle_before_kernel_top=[dict(kernel=k, idle_ms=v[0] / 1e6, gaps=v[1])
                                for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]],
        window_kernel=wk)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
