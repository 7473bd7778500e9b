#!/bin/bash
# round 3, call B: fused kernels after the buffer-store hazard fix — parity, per-launch table fused vs unfused
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "xf_" 2>&1 | tail -8 > gpurun_out/b_xf_tests.txt
cat gpurun_out/b_xf_tests.txt
DBIR_FUSED_XF=1 timeout 600 python tools/profile_eval.py --pair 2>/dev/null | head -45 > gpurun_out/b_profile_fused.txt
DBIR_FUSED_XF=0 timeout 600 python tools/profile_eval.py --pair 2>/dev/null | head -60 > gpurun_out/b_profile_unfused.txt
head -30 gpurun_out/b_profile_fused.txt
