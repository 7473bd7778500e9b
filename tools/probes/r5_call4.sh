#!/bin/bash
# Round-5 GPU call 4: plan replay parity (fixed test), the C5-shape golden against the engine (bf16, 4096 x 4096 x 50 steps),
# deliberate skew between the two encoders (DBIR_ENC_SKEW) in eager and replayed form, tile re-tune at batch 8 / 4.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c4
mkdir -p $O
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -p no:cacheprovider -k "plan_replay or c5_shape" > $O/pipe.log 2>&1
echo "pipeline rc=$? $(tail -1 $O/pipe.log) $(el)"; grep -E "^FAILED|^ERROR|PSNR|Error" $O/pipe.log | head -12
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f img/s  %.1f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null || tail -3 $1; }
for s in -1 0 2 5 -1 8 2 5; do
  timeout 600 env DBIR_ENC_SKEW=$s $B > $O/skew_${s}_$(date +%s).log 2>&1; echo "eager skew $s: $(val $(ls -t $O/skew_${s}_*.log | head -1)) $(el)"
done
for s in -1 2 5; do
  timeout 600 env DBIR_GRAPH=1 DBIR_ENC_SKEW=$s $B > $O/graph_skew_$s.log 2>&1; echo "graph skew $s: $(val $O/graph_skew_$s.log) $(el)"
done
timeout 900 python tools/autotune.py --batch 8 --out $O/tune_b8.json > $O/tune_b8.log 2>&1; echo "autotune b8 rc=$? $(el)"; grep -E "^GEMM launches" $O/tune_b8.log
timeout 900 python tools/autotune.py --batch 4 --out $O/tune_b4.json > $O/tune_b4.log 2>&1; echo "autotune b4 rc=$? $(el)"; grep -E "^GEMM launches" $O/tune_b4.log
