#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 200 python tools/bench_kernels.py --only attn --attn-variants 5,2,6,5,2,6 > gpurun_out/e_attn.log 2>&1; echo "attn rc=$? t=$(( $(date +%s) - T0 ))s"; grep "Lk4096\|Lk1024\|Lk256" gpurun_out/e_attn.log
for v in 2 6; do DBIR_ATTN_VARIANT=$v timeout 200 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "attention" > gpurun_out/e_attn_test$v.log 2>&1; echo "attn tests v$v rc=$? $(tail -1 gpurun_out/e_attn_test$v.log)"; done
for v in 5 2 6 5 2 6; do
DBIR_ATTN_VARIANT=$v timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/e_bench_v$v.log 2>&1
echo "bench v$v rc=$? $(tail -1 gpurun_out/e_bench_v$v.log | cut -c1-120) t=$(( $(date +%s) - T0 ))s"
done
