cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python tools/autotune.py --batch 8 --out gpurun_out/tuning_b8.json > gpurun_out/k_autotune_b8.log 2>&1; echo "autotune b8 rc=$? t=$(( $(date +%s) - T0 ))s $(grep 'GEMM launches' gpurun_out/k_autotune_b8.log)"
timeout 900 python tools/autotune.py --batch 1 --size 4096 --tiled --dtype bf16 --out gpurun_out/tuning_4096.json > gpurun_out/k_autotune_4096.log 2>&1; echo "autotune 4096 rc=$? t=$(( $(date +%s) - T0 ))s $(grep 'GEMM launches' gpurun_out/k_autotune_4096.log)"
python tools/merge_tuning.py gpurun_out/tuning_b8.json --dry | tail -40
cp diffbir_amd/tuning_gfx950.json gpurun_out/k_table_before.json
python tools/merge_tuning.py gpurun_out/tuning_b8.json | tail -1
cp diffbir_amd/tuning_gfx950.json gpurun_out/k_table_after.json
for i in 1 2; do
DBIR_TUNING_FILE=gpurun_out/k_table_before.json timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/k_c2_before$i.log 2>&1; echo "c2 before $(tail -1 gpurun_out/k_c2_before$i.log | cut -c75-115)"
DBIR_TUNING_FILE=gpurun_out/k_table_after.json timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/k_c2_after$i.log 2>&1; echo "c2 after  $(tail -1 gpurun_out/k_c2_after$i.log | cut -c75-115)"
done
echo "t=$(( $(date +%s) - T0 ))s"
