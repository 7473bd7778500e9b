#!/bin/bash
# GPU call B: new tile variants (160-wide, transposed store) -> tests, autotune, full -m gpu suite, bench, rocprof stats.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "glds or splitk or race_screen or linear_transposed" > gpurun_out/b_test_glds.log 2>&1
echo "TEST glds rc=$? : $(tail -1 gpurun_out/b_test_glds.log)  t=$(( $(date +%s) - T0 ))s"
grep -E "^FAILED|^ERROR" gpurun_out/b_test_glds.log | head -20
EX=""
for t in 13 14 15 16; do grep -E "^FAILED" gpurun_out/b_test_glds.log | grep -q "\[.*${t}-\|-${t}-\|\[${t}-" && EX="${EX:+$EX,}$t"; done
echo "exclude='$EX'"
timeout 900 python tools/autotune.py --out gpurun_out/tuning_gfx950.json ${EX:+--exclude $EX} > gpurun_out/b_tune.log 2>&1
echo "autotune rc=$? t=$(( $(date +%s) - T0 ))s"; sed -n 2,2p gpurun_out/b_tune.log
export DBIR_TUNING_FILE=$PWD/gpurun_out/tuning_gfx950.json
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/b_test_all.log 2>&1
echo "TEST all rc=$? : $(tail -1 gpurun_out/b_test_all.log)  t=$(( $(date +%s) - T0 ))s"
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/b_bench.log 2>&1
echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/b_bench.log | cut -c1-2500
