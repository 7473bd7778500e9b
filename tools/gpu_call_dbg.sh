#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -p no:cacheprovider -x -k "tiny or modules" 2>&1 | tail -2
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-330
