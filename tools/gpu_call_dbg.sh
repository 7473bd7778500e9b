#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "glds or splitk or race_screen or linear_transposed or conv3x3 or linear" 2>&1 | tail -3
for P in "conv 16 64 64 320 320 14" "conv 16 64 64 320 320 34" "conv 16 64 64 320 320 37" "conv 16 64 64 320 320 12" "conv 16 64 64 320 320 5" "conv 16 32 32 640 640 14" "conv 16 32 32 640 640 37" "conv 8 128 128 512 512 10" "conv 8 128 128 512 512 30" "conv 8 128 128 512 512 36" "lin 65536 320 320 14" "lin 65536 320 1280 14" "lin 16384 640 640 14" "lin 65536 2560 1280 10"; do
  python tools/bench_one.py $P 20 2>&1 | tail -1
done
echo "== deph instr"; DBIR_GEMM_DEBUG=5 python tools/bench_one.py conv 16 64 64 320 320 37 1 2>&1 | tail -2
