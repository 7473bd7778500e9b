#!/bin/bash
# Round-5 final record at HEAD: smoke(), the default bench line, rocprofv3 kernel statistics of the same command.
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
O=gpurun_out/r5final
mkdir -p $O $O/prof
T0=$(date +%s)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)"
timeout 600 python bench.py --steps 4 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 $O/bench.log | cut -c1-400
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o bench -- python $REPO/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-roofline > $REPO/$O/prof_bench.log 2>&1 )
echo "rocprof rc=$? t=$(( $(date +%s) - T0 ))s"
find $O/prof -name "*kernel_trace.csv" -delete
