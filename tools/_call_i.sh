cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
for b in 3 5 6 7 12 16; do
timeout 600 python tools/autotune.py --batch $b --out gpurun_out/tuning_b$b.json > gpurun_out/i_autotune_b$b.log 2>&1; echo "autotune b$b rc=$? t=$(( $(date +%s) - T0 ))s $(grep 'GEMM launches' gpurun_out/i_autotune_b$b.log)"
done
timeout 900 python tools/autotune.py --batch 1 --size 2048 --tiled --out gpurun_out/tuning_2048.json > gpurun_out/i_autotune_2048.log 2>&1; echo "autotune 2048 rc=$? t=$(( $(date +%s) - T0 ))s $(grep 'GEMM launches' gpurun_out/i_autotune_2048.log)"
