#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider --maxfail=10 -k "window or groupnorm or attention or f32_head" > gpurun_out/r2/c6_tests.log 2>&1
echo "TESTS rc=$? : $(tail -1 gpurun_out/r2/c6_tests.log) $(el)"
grep -E "^FAILED|^ERROR" gpurun_out/r2/c6_tests.log | head -10
timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_multigpu_gpu.py -q -m gpu -p no:cacheprovider --maxfail=10 -k "modules or full_pipeline or gpus_flag or two_ranks" > gpurun_out/r2/c6_pipe.log 2>&1
echo "PIPE rc=$? : $(tail -1 gpurun_out/r2/c6_pipe.log) $(el)"
grep -E "^FAILED|^ERROR" gpurun_out/r2/c6_pipe.log | head -10
for i in 1 2; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r2/c6_bench_new$i.log 2>&1
echo "bench new $i rc=$? $(el) $(tail -1 gpurun_out/r2/c6_bench_new$i.log | cut -c1-130)"
DBIR_TUNING_FILE=$PWD/profiles/tuning_gfx950_r1.json timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r2/c6_bench_r1table$i.log 2>&1
echo "bench r1-table $i rc=$? $(el) $(tail -1 gpurun_out/r2/c6_bench_r1table$i.log | cut -c1-130)"
done
