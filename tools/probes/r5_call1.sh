#!/bin/bash
# Round-5 GPU call 1: new kernels (parity-collapsed upsample convolution, GroupNorm with the statistics merge inside, conv_in
# on the halo / direct-to-LDS kernels), full-size goldens, same-box A/B of each switch, per-shape table, RCCL on one rank,
# GPU-oracle goldens at the benchmarked C3 / C4 shapes.
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
O=gpurun_out/r5c1
mkdir -p $O
T0=$(date +%s)
el() { echo "t=$(( $(date +%s) - T0 ))s"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "up4 or fused_partials or groupnorm or 8p_conv or epilogue_group_norm" > $O/kern.log 2>&1
echo "kernels rc=$? $(tail -1 $O/kern.log) $(el)"; grep -E "^FAILED|^ERROR|Error" $O/kern.log | head -5
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -p no:cacheprovider -k "full_baseline_configs or full_pipeline_50" > $O/pipe.log 2>&1
echo "pipeline rc=$? $(tail -1 $O/pipe.log) $(el)"; grep -E "^FAILED|^ERROR|PSNR" $O/pipe.log | head -12
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
OLD="DBIR_UP4=0 DBIR_CONV_IN_PAD=8 DBIR_GN_FUSED_PARTIALS=0"
val() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f img/s  %.1f ms' % (d['value'], d['ms_per_step']))" 2>/dev/null || tail -2 $1; }
for i in 1 2; do
  timeout 600 $B > $O/ab_new_$i.log 2>&1; echo "A/B new   #$i: $(val $O/ab_new_$i.log) $(el)"
  timeout 600 env $OLD $B > $O/ab_old_$i.log 2>&1; echo "A/B old   #$i: $(val $O/ab_old_$i.log) $(el)"
done
timeout 600 env DBIR_UP4=0 $B > $O/ab_noup4.log 2>&1; echo "A/B UP4=0        : $(val $O/ab_noup4.log) $(el)"
timeout 600 env DBIR_CONV_IN_PAD=8 $B > $O/ab_pad8.log 2>&1; echo "A/B CONV_IN_PAD=8: $(val $O/ab_pad8.log) $(el)"
timeout 600 env DBIR_GN_FUSED_PARTIALS=0 $B > $O/ab_nogn.log 2>&1; echo "A/B GN_FUSED=0   : $(val $O/ab_nogn.log) $(el)"
timeout 600 python tools/profile_eval.py --pair --batch 8 > $O/profile_eval_pair_b8.txt 2>&1; echo "profile_eval rc=$? $(el)"; head -3 $O/profile_eval_pair_b8.txt
timeout 1500 python -m pytest tests/test_multigpu_gpu.py -q -p no:cacheprovider -k "rccl" > $O/rccl.log 2>&1
echo "rccl tests rc=$? $(tail -1 $O/rccl.log) $(el)"; grep -E "^FAILED|^ERROR|Error" $O/rccl.log | head -5
timeout 1500 env MIOPEN_FIND_MODE=FAST python -m oracle.make_golden_gpu > $O/golden_gpu.log 2>&1
echo "gpu oracle rc=$? $(el)"; grep -E "^chain|^case|CHAIN" $O/golden_gpu.log | cut -c1-400
ls -la gpurun_out/golden_gpu 2>/dev/null
