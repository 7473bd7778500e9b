#!/bin/bash
# round 3, call A: fused transformer kernels — parity (phase-by-phase dumps), fetch-path microbenchmark, bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "xf_" -x 2>&1 | tail -25 > gpurun_out/a_xf_tests.txt
cat gpurun_out/a_xf_tests.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/lds_stream_bench.hip -o /tmp/lsb 2>/dev/null && timeout 120 /tmp/lsb > gpurun_out/a_lds_stream.txt 2>&1
cat gpurun_out/a_lds_stream.txt
DBIR_FUSED_XF=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/a_bench_unfused.json.log 2>gpurun_out/a_bench_unfused.err
tail -1 gpurun_out/a_bench_unfused.json.log
DBIR_FUSED_XF=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/a_bench_fused.json.log 2>gpurun_out/a_bench_fused.err
tail -1 gpurun_out/a_bench_fused.json.log; tail -3 gpurun_out/a_bench_fused.err
