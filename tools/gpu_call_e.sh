#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "glds or splitk or race_screen or linear_transposed or conv3x3 or linear or bmm" > gpurun_out/e_test_gemm.log 2>&1
echo "TEST gemm rc=$? : $(tail -1 gpurun_out/e_test_gemm.log)  t=$(( $(date +%s) - T0 ))s"
grep -E "^FAILED|^ERROR" gpurun_out/e_test_gemm.log | head -10
timeout 900 python tools/autotune.py --out gpurun_out/tuning_gfx950.json > gpurun_out/e_tune.log 2>&1
echo "autotune rc=$? t=$(( $(date +%s) - T0 ))s"; sed -n 2,2p gpurun_out/e_tune.log
export DBIR_TUNING_FILE=$PWD/gpurun_out/tuning_gfx950.json
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/e_bench.log 2>&1
echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/e_bench.log | cut -c1-1500
