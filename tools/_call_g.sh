cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
for b in 4 1 2; do
timeout 600 python tools/autotune.py --batch $b --out gpurun_out/tuning_b$b.json > gpurun_out/g_autotune_b$b.log 2>&1; echo "autotune b$b rc=$? t=$(( $(date +%s) - T0 ))s $(grep 'GEMM launches' gpurun_out/g_autotune_b$b.log)"
done
timeout 300 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/g_c3_before.log 2>&1; echo "c3 before rc=$? $(tail -1 gpurun_out/g_c3_before.log | cut -c1-120)"
timeout 300 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/g_b1_before.log 2>&1; echo "b1 before rc=$? $(tail -1 gpurun_out/g_b1_before.log | cut -c1-120)"
