#!/bin/bash
# round 3, call D: cross-tile-prefetch glds tiles (80-88): parity, incremental autotune at the benchmark batch; xf stress
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "glds or xf_" 2>&1 | tail -6 > gpurun_out/d_glds_tests.txt
cat gpurun_out/d_glds_tests.txt
timeout 900 python tools/autotune.py --only 80,81,82,83,84,85,86,87,88 --batch 8 --out gpurun_out/t_xp_b8.json > gpurun_out/d_autotune_xp_b8.log 2>&1
head -50 gpurun_out/d_autotune_xp_b8.log
timeout 300 python bench.py --parity-out gpurun_out/p1.npy --batch 2 --sampler-steps 2 --no-roofline --no-cpu-baseline 2>&1 | tail -2
