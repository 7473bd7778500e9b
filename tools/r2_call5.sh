#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r2
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "attention or f32_head or conv3x3" > gpurun_out/r2/c5_tests.log 2>&1
echo "TESTS rc=$? : $(tail -1 gpurun_out/r2/c5_tests.log) $(el)"
grep -E "^FAILED|^ERROR|Error" gpurun_out/r2/c5_tests.log | head -10
timeout 300 python tools/bench_kernels.py --only attn > gpurun_out/r2/c5_attn.log 2>&1; echo "attn bench rc=$? $(el)"; grep attention gpurun_out/r2/c5_attn.log
timeout 300 python tools/bench_one.py conv 16 64 64 320 4 54 10 2>&1 | tail -1
timeout 300 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -p no:cacheprovider -x -k "samplers or tiny_pipeline" > gpurun_out/r2/c5_pipe.log 2>&1
echo "PIPE rc=$? : $(tail -1 gpurun_out/r2/c5_pipe.log) $(el)"
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2/c5_bench.log 2>&1
echo "bench rc=$? $(el)"; tail -1 gpurun_out/r2/c5_bench.log | cut -c1-300; tail -1 gpurun_out/r2/c5_bench.log | grep -o '"roofline".*' | cut -c1-400
