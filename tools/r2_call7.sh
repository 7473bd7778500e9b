#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 600 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -p no:cacheprovider --maxfail=10 -k "tiled_vae or chunking" > gpurun_out/r2/c7_tests.log 2>&1
echo "TESTS rc=$? : $(tail -1 gpurun_out/r2/c7_tests.log) $(el)"
grep -E "^FAILED|^ERROR|rel_err|enc_tiled" gpurun_out/r2/c7_tests.log | head -10
timeout 600 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2/c7_bench_c4.log 2>&1
echo "bench c4 rc=$? $(el)"; tail -1 gpurun_out/r2/c7_bench_c4.log | cut -c1-900
timeout 900 python bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/r2/c7_bench_c5.log 2>&1
echo "bench c5 rc=$? $(el)"; tail -3 gpurun_out/r2/c7_bench_c5.log | cut -c1-900
python - <<'PY'
import torch
print("peak mem GB", torch.cuda.max_memory_allocated()/2**30)
PY
