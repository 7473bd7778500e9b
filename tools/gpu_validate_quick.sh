#!/bin/bash
# Quick validation on the GPU box (gpurun -- 'bash tools/gpu_validate_quick.sh'): full -m gpu suite, smoke, default bench
# (with cpu baseline), rocprofv3 --kernel-trace --stats of the bench command.  About 8 GPU-minutes.
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$PWD; mkdir -p gpurun_out
T0=$(date +%s)
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/d_test_all.log 2>&1
echo "TEST all rc=$? : $(tail -1 gpurun_out/d_test_all.log)  t=$(( $(date +%s) - T0 ))s"
grep -E "^FAILED|^ERROR" gpurun_out/d_test_all.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/d_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/d_smoke.log)"
timeout 600 python bench.py > gpurun_out/d_bench.log 2>&1
echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/d_bench.log | cut -c1-3000
mkdir -p gpurun_out/prof_d
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_d -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/d_prof_bench.log 2>&1 )
echo "rocprof rc=$? t=$(( $(date +%s) - T0 ))s"
find gpurun_out/prof_d -name "*kernel_trace.csv" -delete
ls gpurun_out/prof_d/* | head
