#!/bin/bash
# Round-2 GPU call 11: persistent-kernel tests (fixed cases), HBM access-pattern diagnostic, autotune of tiles 70-73.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider --maxfail=12 -k "pers" > gpurun_out/r2/c11_tests.log 2>&1
echo "TESTS rc=$? : $(tail -1 gpurun_out/r2/c11_tests.log) $(el)"
grep -E "^FAILED|^ERROR|Error" gpurun_out/r2/c11_tests.log | head -14
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/hbm_pattern.hip -o gpurun_out/r2/hbm_pattern && timeout 120 gpurun_out/r2/hbm_pattern | tee gpurun_out/r2/c11_hbm_pattern.log
rm -f gpurun_out/r2/hbm_pattern
el
timeout 600 python tools/autotune.py --only 70,71,72,73 --out gpurun_out/r2/tuning_pers.json > gpurun_out/r2/c11_tune.log 2>&1
echo "autotune rc=$? $(el)"; head -3 gpurun_out/r2/c11_tune.log | cut -c1-200
grep -E "^0:" gpurun_out/r2/c11_tune.log | cut -c1-300 | head -70
