#!/bin/bash
# round 3, call H: run-to-run determinism / autotune influence on the batch-independence comparison, new tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T0=$(date +%s)
for at in 0 1; do
  echo "== DBIR_AUTOTUNE=$at" >> gpurun_out/h_gn_diag.txt
  DBIR_AUTOTUNE=$at timeout 600 python tools/gn_stats_diag.py tiny full 2>&1 | grep -v amdgpu.ids | grep -v -i warn | tail -8 >> gpurun_out/h_gn_diag.txt
done
cat gpurun_out/h_gn_diag.txt
echo "diag t=$(( $(date +%s) - T0 ))s"
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -k "batch_independence or autotune" 2>&1 | tail -15 > gpurun_out/h_pipeline.log
echo "pipeline tests rc=$? t=$(( $(date +%s) - T0 ))s"; tail -6 gpurun_out/h_pipeline.log
