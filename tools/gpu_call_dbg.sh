#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "attention and not window" 2>&1 | tail -2
python tools/bench_kernels.py --only attn --attn-variants 2 2>&1 | grep attention
