#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "((glds or splitk or linear_transposed) and (40- or 41-)) or (race_screen and (40 or 41))" 2>&1 | tail -3
for P in "conv 8 128 128 512 512 10" "conv 8 128 128 512 512 30" "conv 8 128 128 512 512 40" "conv 8 128 128 512 512 41" "lin 65536 2560 1280 10" "lin 65536 2560 1280 40" "lin 65536 2560 1280 41" "lin 65536 2560 320 30" "lin 65536 2560 320 40" "lin 65536 2560 320 41" "conv 16 16 16 1280 1280 310" "conv 16 16 16 1280 1280 340"; do
  python tools/bench_one.py $P 20 2>&1 | tail -1
done
