#!/bin/bash
# round 3, call M: fused transformer kernels at C = 640: parity again, whole-pipeline golden at batch 8, end-to-end A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "xf_" 2>&1 | tail -6 > gpurun_out/m_kernels.log
echo "kernel tests rc=$? t=$(( $(date +%s) - T0 ))s"; tail -4 gpurun_out/m_kernels.log | cut -c1-220
for rep in 1 2; do
  for v in 320 1; do
    DBIR_FUSED_XF=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/m_bench_xf$v.$rep.log 2>/dev/null
    echo "DBIR_FUSED_XF=$v rep $rep: $(tail -1 gpurun_out/m_bench_xf$v.$rep.log | cut -c1-120)"
  done
done
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -x -k "full and (c2 or fused)" 2>&1 | tail -6 > gpurun_out/m_pipeline.log
echo "pipeline tests rc=$? t=$(( $(date +%s) - T0 ))s"; tail -4 gpurun_out/m_pipeline.log | cut -c1-220
echo "done t=$(( $(date +%s) - T0 ))s"
