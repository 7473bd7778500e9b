#!/bin/bash
# Full validation on the GPU box (run through gpurun from the repo root): -m gpu suite, smoke, bench (with cpu baseline),
# rocprofv3 kernel stats, PMC traffic, the other BASELINE configs (c3 / c4 / c5).  Outputs under gpurun_out/; summaries are
# copied to profiles/ by hand.  tools/gpu_validate_quick.sh = the first four steps only (about 8 GPU-minutes).
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/f_test_all.log 2>&1
echo "TEST all rc=$? : $(tail -1 gpurun_out/f_test_all.log)  t=$(( $(date +%s) - T0 ))s"
grep -E "^FAILED|^ERROR" gpurun_out/f_test_all.log | head
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/f_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/f_smoke.log)"
timeout 900 python bench.py > gpurun_out/f_bench.log 2>&1
echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/f_bench.log | cut -c1-2600
mkdir -p gpurun_out/prof_f
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_f -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/f_prof_bench.log 2>&1 )
echo "rocprof rc=$? t=$(( $(date +%s) - T0 ))s"
find gpurun_out/prof_f -name "*kernel_trace.csv" -delete
sh tools/pmc_traffic.sh gpurun_out/pmc_f > gpurun_out/f_pmc.log 2>&1; echo "pmc rc=$? t=$(( $(date +%s) - T0 ))s"
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/f_$c.log 2>&1; echo "$c rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/f_$c.log | cut -c1-300; done
bash tools/power_probe.sh gpurun_out/power_probe.txt > gpurun_out/f_power.log 2>&1; echo "power probe rc=$? t=$(( $(date +%s) - T0 ))s"
