#!/bin/bash
# round 3, call F: GroupNorm statistics from the producing GEMM's epilogue — kernel tests, goldens, same-box A/B,
# plus the transformer-tail anatomy table DESIGN.md cites
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "epilogue_group or groupnorm or xf_" 2>&1 | tail -15 > gpurun_out/f_kernels.log
echo "kernel tests rc=$? t=$(( $(date +%s) - T0 ))s"; tail -4 gpurun_out/f_kernels.log
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/f_pipeline.log
echo "module/pipeline tests rc=$? t=$(( $(date +%s) - T0 ))s"; tail -4 gpurun_out/f_pipeline.log
for rep in 1 2; do
  for v in 0 1; do
    DBIR_GN_EPILOGUE_STATS=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/f_bench_gn$v.$rep.log 2>/dev/null
    echo "gn_epilogue_stats=$v rep $rep: $(tail -1 gpurun_out/f_bench_gn$v.$rep.log | cut -c1-120)"
  done
done
timeout 300 python tools/xf_anatomy.py 1 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_xf_anatomy.txt; tail -25 gpurun_out/r3_xf_anatomy.txt
echo "done t=$(( $(date +%s) - T0 ))s"
