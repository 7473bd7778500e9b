#!/bin/bash
# Round-5 GPU call 9: HIP stream priority of the ControlNet's stream (eager) / of the plan's side stream (replay).
# (The two knobs DBIR_SIDE_PRIO / DBIR_PLAN_SIDE_PRIO existed at commit d8ea50b+ only; they measured null / -18 % and were removed:
#  profiles/r5_stream_priority_ab.txt.  Kept as the record of what was run.)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c9
mkdir -p $O
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
python -c "import torch; print('torch priority range', torch.cuda.Stream.priority_range())"
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f img/s  %.1f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null || tail -3 $1; }
for p in 0 -1 0 -1 1; do
  timeout 600 env DBIR_SIDE_PRIO=$p $B > $O/eager_prio_${p}_$(date +%s).log 2>&1; echo "eager side prio $p: $(val $(ls -t $O/eager_prio_${p}_*.log | head -1)) $(el)"
done
for p in 0 -1 1; do
  timeout 600 env DBIR_GRAPH=1 DBIR_PLAN=1 DBIR_PLAN_SIDE_PRIO=$p $B > $O/plan_prio_$p.log 2>&1; echo "plan side prio $p: $(val $O/plan_prio_$p.log) $(el)"
done
