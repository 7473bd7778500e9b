#!/bin/bash
# GPU call A: validate the new kernel variants, autotune tiles in situ, A/B the attention kernels, short bench.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
T0=$(date +%s)
run_tests() {  # name, -k expression
  timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "$2" -p no:cacheprovider > gpurun_out/a_test_$1.log 2>&1
  echo "TEST $1 rc=$? : $(tail -1 gpurun_out/a_test_$1.log)"
}
run_tests ph "race_screen or (glds and 13-)"
run_tests splitk "glds_splitk"
run_tests attn "attention and not window"
run_tests tiles "sampler_and_tiles"
EX=""
grep -q "failed" gpurun_out/a_test_ph.log && EX="13"
grep -q "failed" gpurun_out/a_test_splitk.log && EX="${EX:+$EX,}splitk"
if grep -q "attn_v2" gpurun_out/a_test_attn.log && grep -q "failed" gpurun_out/a_test_attn.log; then export DBIR_ATTN_VARIANT=1; fi
echo "exclude='$EX' attn_variant=${DBIR_ATTN_VARIANT:-2}  t=$(( $(date +%s) - T0 ))s"
timeout 900 python tools/autotune.py --out gpurun_out/tuning_gfx950.json ${EX:+--exclude $EX} > gpurun_out/a_tune.log 2>&1
echo "autotune rc=$? t=$(( $(date +%s) - T0 ))s"; head -3 gpurun_out/a_tune.log
timeout 300 python tools/bench_kernels.py --only attn --attn-variants 1,2 > gpurun_out/a_attn.log 2>&1
echo "attn bench rc=$?"; cat gpurun_out/a_attn.log | tail -12
DBIR_TUNING_FILE=$PWD/gpurun_out/tuning_gfx950.json timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/a_bench.log 2>&1
echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/a_bench.log | cut -c1-1500
