#!/bin/bash
# Round-5 GPU call 10: persistent context K / V^T buffer sets (replays survive a new prompt tensor): parity + A/B.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c10
mkdir -p $O
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -p no:cacheprovider -k "plan or refresh_the_context or tiny_pipeline_vs_reference or graph" > $O/pipe.log 2>&1
echo "pipeline rc=$? $(tail -1 $O/pipe.log) $(el)"; grep -E "^FAILED|^ERROR|Error" $O/pipe.log | head -8
B="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f img/s  %.1f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null || tail -3 $1; }
for i in 1 2; do
  timeout 600 env DBIR_GRAPH=0 $B > $O/b8_eager_$i.log 2>&1; echo "b8 eager #$i: $(val $O/b8_eager_$i.log) $(el)"
  timeout 600 env DBIR_GRAPH=1 $B > $O/b8_graph_$i.log 2>&1; echo "b8 graph #$i: $(val $O/b8_graph_$i.log) $(el)"
  timeout 600 env DBIR_GRAPH=1 DBIR_PLAN=1 $B > $O/b8_plan_$i.log 2>&1; echo "b8 plan  #$i: $(val $O/b8_plan_$i.log) $(el)"
done
B1="$B --batch 1"
timeout 600 env DBIR_GRAPH=0 $B1 > $O/b1_eager.log 2>&1; echo "b1 eager: $(val $O/b1_eager.log) $(el)"
timeout 600 $B1 > $O/b1_graph.log 2>&1; echo "b1 graph (auto): $(val $O/b1_graph.log) $(el)"
timeout 600 env DBIR_PLAN=1 $B1 > $O/b1_plan.log 2>&1; echo "b1 plan (auto) : $(val $O/b1_plan.log) $(el)"
