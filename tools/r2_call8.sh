#!/bin/bash
# Round-2 GPU call 8: 4-weight-slot lockstep halo variants (tiles 55/56): tests, autotune vs the current table.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 420 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider --maxfail=10 -k "halo" > gpurun_out/r2/c8_tests.log 2>&1
echo "TESTS rc=$? : $(tail -1 gpurun_out/r2/c8_tests.log) $(el)"
grep -E "^FAILED|^ERROR|Error" gpurun_out/r2/c8_tests.log | head -10
timeout 600 python tools/autotune.py --only 55,56 --out gpurun_out/r2/tuning_halo4.json > gpurun_out/r2/c8_tune.log 2>&1
echo "autotune rc=$? $(el)"; head -3 gpurun_out/r2/c8_tune.log | cut -c1-200
grep -E "^1:" gpurun_out/r2/c8_tune.log | cut -c1-420 | head -60
