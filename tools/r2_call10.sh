#!/bin/bash
# Round-2 GPU call 10: persistent linear kernel (tiles 70-73): tests, isolated A/B on the transformer shapes.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider --maxfail=12 -k "pers" > gpurun_out/r2/c10_tests.log 2>&1
echo "TESTS rc=$? : $(tail -1 gpurun_out/r2/c10_tests.log) $(el)"
grep -E "^FAILED|^ERROR|Error" gpurun_out/r2/c10_tests.log | head -14
for spec in "65536 320 320 14" "65536 320 320 70" "65536 320 320 71" "65536 320 320 72" "65536 320 320 73" \
            "65536 640 320 44" "65536 640 320 70" "65536 640 320 71" "65536 640 320 73" \
            "65536 320 1280 37" "65536 320 1280 70" "65536 320 1280 71" \
            "16384 640 640 37" "16384 640 640 70" "16384 640 640 71" "16384 640 640 72" "16384 640 640 73" \
            "16384 1280 640 35" "16384 1280 640 70" "16384 1280 640 71" "16384 640 2560 37" "16384 640 2560 70" "16384 640 2560 71" \
            "4096 1280 1280 25" "4096 1280 1280 70" "4096 1280 1280 71" "4096 1280 1280 72" "4096 1280 1280 73" \
            "4096 1280 5120 25" "4096 1280 5120 70" "4096 1280 5120 71" "4096 2560 1280 37" "4096 2560 1280 70" "4096 2560 1280 71"; do
  timeout 120 python tools/bench_one.py lin $spec 20 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r2/c10_ours.log
echo "dephase 0:"
for spec in "65536 320 320 71" "65536 320 320 73" "16384 640 640 71" "4096 1280 1280 71"; do
  DBIR_PERS_DEPHASE=0 timeout 120 python tools/bench_one.py lin $spec 20 2>&1 | grep -v amdgpu.ids
done | tee -a gpurun_out/r2/c10_ours.log
el
