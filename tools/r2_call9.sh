#!/bin/bash
# Round-2 GPU call 9: vendor-library yardstick on the small-K linear shapes vs our tiles, back-to-back timing.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2
timeout 300 python tools/blas_yardstick.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/c9_blas.log
for spec in "65536 320 320 14" "65536 320 320 15" "65536 320 320 45" "65536 320 320 35" "65536 640 320 44" "65536 320 1280 37" "16384 640 640 37" "16384 640 640 34" "16384 640 640 45" "16384 1280 640 35" "16384 640 2560 37" "4096 1280 1280 25" "4096 1280 1280 45" "4096 2560 1280 37" "4096 1280 5120 25" "1024 1280 1280 325"; do
  timeout 120 python tools/bench_one.py lin $spec 20 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r2/c9_ours.log
