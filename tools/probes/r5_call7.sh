#!/bin/bash
# Round-5 GPU call 7: does the RATE at which a replay enqueues its launches matter?  (plan replay with a busy-wait between
# recorded calls: DBIR_PLAN_PACE_NS) + the refreshed tile table against the round-4 table in situ
# (before the call: git show 8e4ae0a:diffbir_amd/tuning_gfx950.json > gpurun_tuning_r4.json   — git-ignored scratch copy).
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c7
mkdir -p $O
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f img/s  %.1f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null || tail -3 $1; }
timeout 600 $B > $O/eager_1.log 2>&1; echo "eager            : $(val $O/eager_1.log) $(el)"
for p in 0 5000 15000 30000; do
  timeout 600 env DBIR_GRAPH=1 DBIR_PLAN=1 DBIR_PLAN_PACE_NS=$p $B > $O/plan_pace_$p.log 2>&1; echo "plan pace ${p} ns: $(val $O/plan_pace_$p.log) $(el)"
done
timeout 600 $B > $O/eager_2.log 2>&1; echo "eager            : $(val $O/eager_2.log) $(el)"
timeout 600 env DBIR_TUNING_FILE=$PWD/gpurun_tuning_r4.json $B > $O/table_r4_1.log 2>&1; echo "eager, r4 table : $(val $O/table_r4_1.log) $(el)"
timeout 600 $B > $O/eager_3.log 2>&1; echo "eager            : $(val $O/eager_3.log) $(el)"
timeout 600 env DBIR_TUNING_FILE=$PWD/gpurun_tuning_r4.json $B > $O/table_r4_2.log 2>&1; echo "eager, r4 table : $(val $O/table_r4_2.log) $(el)"
