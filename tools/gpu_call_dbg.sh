#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for P in "conv 16 64 64 320 320 14" "conv 8 128 128 512 512 10" "conv 16 64 64 320 320 12"; do
  for D in 0 1 3; do
    echo -n "debug=$D  "; DBIR_GEMM_DEBUG=$D python tools/bench_one.py $P 20 2>&1 | tail -1
  done
done
