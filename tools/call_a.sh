#!/bin/bash
# round-2 re-entry call A: new kernels + pipeline parity + A/B of the shared CFG prefix
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "clip or add_layernorm or causal" > gpurun_out/a_kern.log 2>&1
echo "kern rc=$? $(tail -1 gpurun_out/a_kern.log) t=$(( $(date +%s) - T0 ))s"
timeout 600 python -m pytest tests/test_pipeline_gpu.py -q -p no:cacheprovider -x > gpurun_out/a_pipe.log 2>&1
echo "pipe rc=$? $(tail -1 gpurun_out/a_pipe.log) t=$(( $(date +%s) - T0 ))s"; grep -E "^FAILED|^ERROR" gpurun_out/a_pipe.log | head
DBIR_SHARE_CFG_PREFIX=0 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/a_bench_noshare.log 2>&1
echo "bench noshare rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/a_bench_noshare.log | cut -c1-330
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/a_bench_share.log 2>&1
echo "bench share rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/a_bench_share.log | cut -c1-330
