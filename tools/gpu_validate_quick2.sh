#!/bin/bash
# After a change confined to one kernel family: that family's kernel tests + every pipeline golden + smoke + the default bench
# + the rocprofv3 kernel statistics (about 4 GPU-minutes).  Usage: bash tools/gpu_validate_quick2.sh "<pytest -k expression>"
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
mkdir -p gpurun_out
T0=$(date +%s)
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "$1" > gpurun_out/q_kern.log 2>&1; echo "kernels rc=$? $(tail -1 gpurun_out/q_kern.log)"
timeout 600 python -m pytest tests/test_pipeline_gpu.py -q -p no:cacheprovider > gpurun_out/q_pipe.log 2>&1; echo "pipeline rc=$? $(tail -1 gpurun_out/q_pipe.log) t=$(( $(date +%s) - T0 ))s"
grep -E "^FAILED|^ERROR" gpurun_out/q_pipe.log gpurun_out/q_kern.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/f_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/f_smoke.log)"
timeout 900 python bench.py > gpurun_out/f_bench.log 2>&1; echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/f_bench.log | cut -c1-300
mkdir -p gpurun_out/prof_f
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_f -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/f_prof_bench.log 2>&1 )
echo "rocprof rc=$? t=$(( $(date +%s) - T0 ))s"
find gpurun_out/prof_f -name "*kernel_trace.csv" -delete
