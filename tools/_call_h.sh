cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests/test_pipeline_gpu.py -q -p no:cacheprovider -x -k "graph_replay or tiny_pipeline or cleaner_pipelines" > gpurun_out/h_pipe.log 2>&1
echo "pipe rc=$? $(tail -1 gpurun_out/h_pipe.log) t=$(( $(date +%s) - T0 ))s"; grep -E "^FAILED|^ERROR|Error" gpurun_out/h_pipe.log | head
for g in 0 1 auto; do
DBIR_GRAPH=$g timeout 300 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/h_c3_g$g.log 2>&1; echo "c3 graph=$g rc=$? $(tail -1 gpurun_out/h_c3_g$g.log | cut -c1-100)"
done
for g in 0 1; do
DBIR_GRAPH=$g timeout 300 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/h_b1_g$g.log 2>&1; echo "b1 graph=$g rc=$? $(tail -1 gpurun_out/h_b1_g$g.log | cut -c1-120)"
done
for g in 0 1 auto; do
DBIR_GRAPH=$g timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/h_c2_g$g.log 2>&1; echo "c2 graph=$g rc=$? $(tail -1 gpurun_out/h_c2_g$g.log | cut -c1-120) t=$(( $(date +%s) - T0 ))s"
done
