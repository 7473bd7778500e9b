#!/bin/bash
# call C: attention occupancy A/B (variants 2 / 4 / 5), then PMC traffic of the GEMM family with the round-2 kernels
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 200 python tools/bench_kernels.py --only attn --attn-variants 2,4,5,2,4,5 > gpurun_out/c_attn.log 2>&1; echo "attn rc=$? t=$(( $(date +%s) - T0 ))s"; grep "Lk4096\|Lk1024\|Lk256" gpurun_out/c_attn.log
DBIR_ATTN_VARIANT=4 timeout 200 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "attention" > gpurun_out/c_attn_test4.log 2>&1; echo "attn tests v4 rc=$? $(tail -1 gpurun_out/c_attn_test4.log)"
for v in 2 4 5 2 4; do
DBIR_ATTN_VARIANT=$v timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/c_bench_v$v.log 2>&1
echo "bench v$v rc=$? $(tail -1 gpurun_out/c_bench_v$v.log | cut -c1-120) t=$(( $(date +%s) - T0 ))s"
done
sh tools/pmc_traffic.sh gpurun_out/pmc_c > gpurun_out/c_pmc.log 2>&1; echo "pmc rc=$? t=$(( $(date +%s) - T0 ))s"; tail -5 gpurun_out/c_pmc.log
