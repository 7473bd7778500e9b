#!/bin/bash
# Round-5 GPU call 3: dbir_plan (recorded evaluation replayed from C) parity + A/B against eager launches and HIP-graph replay at
# batch 8 and batch 1; engine vs the GPU-oracle goldens of C3 b4 / C4 x 50 steps; the C5-shape golden (fp32 oracle on the GPU).
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c3
mkdir -p $O
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -p no:cacheprovider -k "plan or (gpu_oracle_golden and not c5)" > $O/pipe.log 2>&1
echo "pipeline rc=$? $(tail -1 $O/pipe.log) $(el)"; grep -E "^FAILED|^ERROR|PSNR|Error" $O/pipe.log | head -12
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f img/s  %.1f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null || tail -3 $1; }
for i in 1 2; do
  timeout 600 env DBIR_GRAPH=0 $B > $O/b8_eager_$i.log 2>&1; echo "b8 eager #$i: $(val $O/b8_eager_$i.log) $(el)"
  timeout 600 env DBIR_GRAPH=1 $B > $O/b8_graph_$i.log 2>&1; echo "b8 graph #$i: $(val $O/b8_graph_$i.log) $(el)"
  timeout 600 env DBIR_GRAPH=1 DBIR_PLAN=1 $B > $O/b8_plan_$i.log 2>&1; echo "b8 plan  #$i: $(val $O/b8_plan_$i.log) $(el)"
done
B1="$B --batch 1"
timeout 600 env DBIR_GRAPH=0 $B1 > $O/b1_eager.log 2>&1; echo "b1 eager: $(val $O/b1_eager.log) $(el)"
timeout 600 env DBIR_GRAPH=1 $B1 > $O/b1_graph.log 2>&1; echo "b1 graph: $(val $O/b1_graph.log) $(el)"
timeout 600 env DBIR_GRAPH=1 DBIR_PLAN=1 $B1 > $O/b1_plan.log 2>&1; echo "b1 plan : $(val $O/b1_plan.log) $(el)"
timeout 900 env MIOPEN_FIND_MODE=FAST python -m oracle.make_golden_gpu c5 > $O/golden_c5.log 2>&1
echo "gpu oracle c5 rc=$? $(el)"; grep -E "^case|Error|error|saved" $O/golden_c5.log | cut -c1-300 | head -5
ls -la gpurun_out/golden_gpu 2>/dev/null
