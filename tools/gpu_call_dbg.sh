#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "((glds or splitk or linear_transposed) and (36- or 37- or 38-)) or race_screen" 2>&1 | tail -4
for P in "conv 16 64 64 320 320 14" "conv 16 64 64 320 320 37" "conv 16 64 64 320 320 12" "conv 16 64 64 320 320 36" "conv 16 32 32 640 640 14" "conv 16 32 32 640 640 37" "conv 8 128 128 512 512 10" "conv 8 128 128 512 512 36" "lin 65536 320 320 37" "lin 65536 320 320 14" "lin 65536 320 1280 14" "lin 65536 320 1280 37" "lin 16384 640 640 37" "lin 16384 640 640 14"; do
  python tools/bench_one.py $P 20 2>&1 | tail -1
done
