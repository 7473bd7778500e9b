#!/bin/bash
# round 3, call G: epilogue GroupNorm statistics — all tile variants, perturbation diagnostic, per-shape table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "epilogue_group" 2>&1 | tail -25 > gpurun_out/g_kernels.log
echo "kernel tests rc=$? t=$(( $(date +%s) - T0 ))s"; tail -12 gpurun_out/g_kernels.log
timeout 600 python tools/gn_stats_diag.py tiny full 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/g_gn_diag.txt; cat gpurun_out/g_gn_diag.txt
echo "diag t=$(( $(date +%s) - T0 ))s"
timeout 300 python tools/profile_eval.py --pair --batch 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_profile_eval_pair_b8.txt; head -60 gpurun_out/r3_profile_eval_pair_b8.txt | cut -c1-200
echo "done t=$(( $(date +%s) - T0 ))s"
