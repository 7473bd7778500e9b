#!/bin/bash
# Round-5 GPU call 8: rocprofv3 kernel statistics of the same pass under eager launches and under plan replay (which kernels
# take longer when the identical launch sequence is replayed from C?)
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
O=$REPO/gpurun_out/r5c8
mkdir -p $O/eager $O/plan
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eager -o p -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/eager.log 2>&1; echo "eager rc=$? $(tail -1 $O/eager.log | cut -c1-120)"
DBIR_GRAPH=1 DBIR_PLAN=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/plan -o p -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/plan.log 2>&1; echo "plan rc=$? $(tail -1 $O/plan.log | cut -c1-120)"
cd $REPO
python tools/idle_gaps.py $O/eager > $O/idle_eager.json 2>/dev/null; python tools/idle_gaps.py $O/plan > $O/idle_plan.json 2>/dev/null
find $O -name "*kernel_trace.csv" -delete
ls $O/eager $O/plan
