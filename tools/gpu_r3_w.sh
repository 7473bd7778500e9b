#!/bin/bash
# round 3, call W: validation of the final tree + per-config lines + HBM traffic of the final kernels
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$PWD
mkdir -p gpurun_out
T0=$(date +%s)
bash tools/gpu_validate_quick.sh
echo "validate done t=$(( $(date +%s) - T0 ))s"
sh tools/pmc_traffic.sh gpurun_out/pmc_b8 8 > gpurun_out/w_pmc_b8.log 2>&1; echo "pmc b8 rc=$? t=$(( $(date +%s) - T0 ))s"
cp gpurun_out/pmc_b8/pmc_traffic.json gpurun_out/r3_pmc_traffic_b8_final.json 2>/dev/null
rm -rf gpurun_out/pmc_b8/FETCH_SIZE gpurun_out/pmc_b8/WRITE_SIZE
for c in c3 c4; do timeout 600 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/w_bench_$c.json.log 2>/dev/null; echo "$c rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/w_bench_$c.json.log | cut -c1-300; done
