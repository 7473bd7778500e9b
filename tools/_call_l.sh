cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/l_c5.log 2>&1; echo "c5 rc=$? t=$(( $(date +%s) - T0 ))s $(tail -1 gpurun_out/l_c5.log | cut -c1-110)"
timeout 300 python -m pytest tests/test_pipeline_gpu.py -q -p no:cacheprovider -x -k "graph_replay" > gpurun_out/l_graph.log 2>&1; echo "graph test rc=$? $(tail -1 gpurun_out/l_graph.log)"
