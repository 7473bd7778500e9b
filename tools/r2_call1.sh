#!/bin/bash
# Round-2 GPU call 1: validate graph capture + halo kernel, measure idle gaps, tune the new tiles.
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
mkdir -p gpurun_out/r2
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --maxfail=25 > gpurun_out/r2/c1_tests.log 2>&1
echo "TESTS rc=$? : $(tail -1 gpurun_out/r2/c1_tests.log) $(el)"
grep -E "^FAILED|^ERROR" gpurun_out/r2/c1_tests.log | head -30
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2/c1_bench_graph.log 2>&1
echo "bench(graph) rc=$? $(el)"; tail -1 gpurun_out/r2/c1_bench_graph.log | cut -c1-700
DBIR_GRAPH=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r2/c1_bench_eager.log 2>&1
echo "bench(eager) rc=$? $(el)"; tail -1 gpurun_out/r2/c1_bench_eager.log | cut -c1-400
for mode in 0 1; do
  mkdir -p gpurun_out/r2/trace$mode
  ( cd /tmp; export TMPDIR=/tmp; DBIR_GRAPH=$mode timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/r2/trace$mode -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/r2/c1_trace$mode.log 2>&1 )
  echo "trace graph=$mode rc=$? $(el)"
  python tools/idle_gaps.py gpurun_out/r2/trace$mode --window-kernel spaced_step > gpurun_out/r2/c1_idle_gaps_graph$mode.json 2>&1
  head -12 gpurun_out/r2/c1_idle_gaps_graph$mode.json
  find gpurun_out/r2/trace$mode -name "*.csv" -size +20M -delete
done
timeout 900 python tools/autotune.py --only 50,51 --out gpurun_out/r2/tuning_halo.json > gpurun_out/r2/c1_tune.log 2>&1
echo "autotune rc=$? $(el)"; head -3 gpurun_out/r2/c1_tune.log | cut -c1-200
grep -E "^1:" gpurun_out/r2/c1_tune.log | cut -c1-330 | head -40
