#!/bin/bash
# Round-2 GPU call 2: halo ablations, 2-blocks-per-CU BK32 variants (autotune), kernel stats of the bench with halo tiles.
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
mkdir -p gpurun_out/r2
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "halo or glds_linear or glds_geglu or glds_conv3x3" > gpurun_out/r2/c2_tests.log 2>&1
echo "TESTS rc=$? : $(tail -1 gpurun_out/r2/c2_tests.log) $(el)"
timeout 300 python tools/halo_ablate.py > gpurun_out/r2/c2_ablate.log 2>&1; echo "ablate rc=$? $(el)"; cat gpurun_out/r2/c2_ablate.log | grep -v amdgpu.ids
timeout 600 python tools/autotune.py --only 44,45 --out gpurun_out/r2/tuning_bk32.json > gpurun_out/r2/c2_tune.log 2>&1
echo "autotune rc=$? $(el)"; head -3 gpurun_out/r2/c2_tune.log | cut -c1-200
grep -E "^0:" gpurun_out/r2/c2_tune.log | cut -c1-200 | head -50
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2/c2_bench.log 2>&1
echo "bench rc=$? $(el)"; tail -1 gpurun_out/r2/c2_bench.log | cut -c1-1500
mkdir -p gpurun_out/r2/stats
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r2/stats -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/r2/c2_stats_run.log 2>&1 )
echo "rocprof stats rc=$? $(el)"
find gpurun_out/r2/stats -name "*kernel_trace.csv" -delete
find gpurun_out/r2/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {} | cut -c1-160'
