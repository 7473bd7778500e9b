#!/bin/bash
# Round-5 GPU call 2: engine vs the new GPU-oracle goldens (C3 b4, C4 x 50 steps), HIP-graph replay A/B at batch 8 under the
# two-stream schedule (VERDICT r4 #5 / DESIGN r4 §9.0), the C5-shape golden (4096 x 4096 x 50 steps, fp32 oracle on the GPU).
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c2
mkdir -p $O
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -p no:cacheprovider -k "gpu_oracle_golden and not c5" > $O/pipe.log 2>&1
echo "pipeline rc=$? $(tail -1 $O/pipe.log) $(el)"; grep -E "^FAILED|^ERROR|PSNR" $O/pipe.log | head -12
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f img/s  %.1f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null || tail -2 $1; }
for i in 1 2; do
  timeout 600 env DBIR_GRAPH=0 $B > $O/graph0_$i.log 2>&1; echo "eager (DBIR_GRAPH=0) #$i: $(val $O/graph0_$i.log) $(el)"
  timeout 600 env DBIR_GRAPH=1 $B > $O/graph1_$i.log 2>&1; echo "graph (DBIR_GRAPH=1) #$i: $(val $O/graph1_$i.log) $(el)"
done
timeout 900 env MIOPEN_FIND_MODE=FAST python -m oracle.make_golden_gpu c5 > $O/golden_c5.log 2>&1
echo "gpu oracle c5 rc=$? $(el)"; grep -E "^case|Error|error" $O/golden_c5.log | cut -c1-300
ls -la gpurun_out/golden_gpu 2>/dev/null
