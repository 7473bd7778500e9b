#!/bin/bash
# Round-2 GPU call 3: lockstep-prefetch halo schedule (tiles 52/53): tests, ablations, autotune vs incumbents.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "halo" > gpurun_out/r2/c3_tests.log 2>&1
echo "TESTS rc=$? : $(tail -1 gpurun_out/r2/c3_tests.log) $(el)"
grep -E "^FAILED|^ERROR|Error" gpurun_out/r2/c3_tests.log | head -10
timeout 300 python tools/halo_ablate.py > gpurun_out/r2/c3_ablate.log 2>&1; echo "ablate rc=$? $(el)"; cat gpurun_out/r2/c3_ablate.log | grep -v amdgpu.ids
timeout 600 python tools/autotune.py --only 50,51,52,53 --out gpurun_out/r2/tuning_halo_ls.json > gpurun_out/r2/c3_tune.log 2>&1
echo "autotune rc=$? $(el)"; head -3 gpurun_out/r2/c3_tune.log | cut -c1-200
grep -E "^1:" gpurun_out/r2/c3_tune.log | cut -c1-420 | head -45
