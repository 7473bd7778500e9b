#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "glds or splitk or race_screen or linear_transposed or conv3x3 or linear" > gpurun_out/d_test_glds.log 2>&1
echo "TEST gemm rc=$? : $(tail -1 gpurun_out/d_test_glds.log)  t=$(( $(date +%s) - T0 ))s"
grep -E "^FAILED|^ERROR" gpurun_out/d_test_glds.log | head -10
for P in "conv 16 64 64 320 320 14" "conv 8 128 128 512 512 10" "conv 8 128 128 512 512 13" "conv 16 16 16 1280 1280 12" "conv 16 32 32 640 640 14" "conv 8 256 256 256 256 10"; do
  for TI in 0 1; do echo -n "tap_inner=$TI  "; DBIR_TAP_INNER=$TI python tools/bench_one.py $P 20 2>&1 | tail -1; done
done
timeout 900 python tools/autotune.py --out gpurun_out/tuning_gfx950.json > gpurun_out/d_tune.log 2>&1
echo "autotune rc=$? t=$(( $(date +%s) - T0 ))s"; sed -n 2,2p gpurun_out/d_tune.log
export DBIR_TUNING_FILE=$PWD/gpurun_out/tuning_gfx950.json
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/d_bench.log 2>&1
echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/d_bench.log | cut -c1-1300
