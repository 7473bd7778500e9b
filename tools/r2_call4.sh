#!/bin/bash
# Round-2 GPU call 4: full -m gpu suite (new samplers, full-config goldens), smoke, bench c2 (+cpu baseline) and c3, kernel stats.
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
mkdir -p gpurun_out/r2
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --maxfail=25 > gpurun_out/r2/c4_tests.log 2>&1
echo "TESTS rc=$? : $(tail -1 gpurun_out/r2/c4_tests.log) $(el)"
grep -E "^FAILED|^ERROR" gpurun_out/r2/c4_tests.log | head -30
cp gpurun_out/pipeline_report.json gpurun_out/r2/c4_pipeline_report.json 2>/dev/null
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2/c4_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/r2/c4_smoke.log) $(el)"
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r2/c4_bench_c2.log 2>&1
echo "bench c2 rc=$? $(el)"; tail -1 gpurun_out/r2/c4_bench_c2.log | cut -c1-2400
timeout 600 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2/c4_bench_c3.log 2>&1
echo "bench c3 rc=$? $(el)"; tail -1 gpurun_out/r2/c4_bench_c3.log | cut -c1-1200
mkdir -p gpurun_out/r2/stats4
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r2/stats4 -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/r2/c4_stats_run.log 2>&1 )
echo "rocprof stats rc=$? $(el)"
find gpurun_out/r2/stats4 -name "*kernel_trace.csv" -delete
find gpurun_out/r2/stats4 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -16 {} | cut -c1-150'
