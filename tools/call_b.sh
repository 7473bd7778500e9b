#!/bin/bash
# round-2 re-entry call B: cleaners (N3) + CLIP (N4) + shared prefix on the GPU; bench with roofline
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "block2x2 or window_attention or clip or add_layernorm or causal" > gpurun_out/b_kern.log 2>&1
echo "kern rc=$? $(tail -1 gpurun_out/b_kern.log) t=$(( $(date +%s) - T0 ))s"; grep -E "^FAILED|^ERROR" gpurun_out/b_kern.log | head
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -p no:cacheprovider -k "cleaner or tiny_pipeline or options" > gpurun_out/b_pipe.log 2>&1
echo "pipe rc=$? $(tail -1 gpurun_out/b_pipe.log) t=$(( $(date +%s) - T0 ))s"; grep -E "^FAILED|^ERROR" gpurun_out/b_pipe.log | head -20
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_bench.log 2>&1
echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/b_bench.log | cut -c1-1800
