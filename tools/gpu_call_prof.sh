#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command (1 timed pass) -> gpurun_out/prof_*; summary copied to profiles/ by hand
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
mkdir -p gpurun_out/prof
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/prof_bench.log 2>&1
echo "rc=$?"; tail -1 $REPO/gpurun_out/prof_bench.log | cut -c1-400
find $REPO/gpurun_out/prof -name "*kernel_stats.csv" | head -3
# keep only the stats (the raw trace is large)
find $REPO/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
