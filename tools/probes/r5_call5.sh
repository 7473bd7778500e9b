#!/bin/bash
# Round-5 GPU call 5: what do the time-embedding GEMMs at the head of each stream cost when they are not cached (as in a replay)?
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c5
mkdir -p $O
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f img/s  %.1f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null || tail -3 $1; }
for i in 1 2; do
  timeout 600 $B > $O/cache1_$i.log 2>&1; echo "eager, cache on  #$i: $(val $O/cache1_$i.log) $(el)"
  timeout 600 env DBIR_TEMB_CACHE=0 $B > $O/cache0_$i.log 2>&1; echo "eager, cache off #$i: $(val $O/cache0_$i.log) $(el)"
done
timeout 600 env DBIR_TEMB_CACHE=0 $B --batch 1 > $O/b1_cache0.log 2>&1; echo "b1 eager, cache off: $(val $O/b1_cache0.log) $(el)"
timeout 600 env DBIR_GRAPH=0 $B --batch 1 > $O/b1_cache1.log 2>&1; echo "b1 eager, cache on : $(val $O/b1_cache1.log) $(el)"
