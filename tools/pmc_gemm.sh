#!/bin/bash
# SQ / TCC counters for a few GEMM problems (one kernel per run): where do the waves wait?
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
OUT=$REPO/gpurun_out/pmcg
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
# round 2: the halo-patch convolution (50 de-phased / 52 lockstep), the persistent linear tiles and the small-M floor
declare -a PROBS=("conv 16 64 64 320 320 50" "conv 16 64 64 320 320 52" "conv 16 32 32 640 640 50" "lin 65536 320 320 71" "lin 65536 320 1280 37" "lin 4096 1280 1280 25")
declare -a SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum")
pi=0
for P in "${PROBS[@]}"; do
  si=0
  for S in "${SETS[@]}"; do
    timeout 120 rocprofv3 --pmc $S --kernel-trace --kernel-include-regex "gemm" --output-format csv -d $OUT/p${pi}_s${si} -o r -- python $REPO/tools/probes/bench_one.py $P 3 > $OUT/p${pi}_s${si}.log 2>&1 || echo "FAILED set $si for $P: $(tail -2 $OUT/p${pi}_s${si}.log | head -1 | cut -c1-200)"
    si=$((si+1))
  done
  grep "us/launch" $OUT/p${pi}_s0.log
  pi=$((pi+1))
done
python - <<PY
import csv, glob, os, collections
out="$OUT"
for d in sorted(glob.glob(out+"/p*_s*")):
    if not os.path.isdir(d): continue
    agg=collections.defaultdict(lambda: [0,0.0])
    for fn in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k=(r["Kernel_Name"][:60], r["Counter_Name"])
            agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    for (kn,cn),(n,v) in sorted(agg.items()):
        print(os.path.basename(d), kn, cn, n, f"{v/n:.4g}")
PY
find $OUT -name "*kernel_trace.csv" -delete
