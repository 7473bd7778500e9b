#!/bin/bash
# round 3, call E: per-config bench lines + rocprof stats, PMC traffic per evaluation batch, rocprof-vs-wall, autotune-on-miss
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$PWD
mkdir -p gpurun_out
T0=$(date +%s)
for B in 8 4 16; do
  sh tools/pmc_traffic.sh gpurun_out/pmc_b$B $B > gpurun_out/e_pmc_b$B.log 2>&1; echo "pmc b$B rc=$? t=$(( $(date +%s) - T0 ))s"
  cp gpurun_out/pmc_b$B/pmc_traffic.json gpurun_out/r3_pmc_traffic_b$B.json 2>/dev/null
  rm -rf gpurun_out/pmc_b$B/FETCH_SIZE gpurun_out/pmc_b$B/WRITE_SIZE
done
timeout 300 python tools/autotune_miss_check.py 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/e_autotune_miss.txt; cat gpurun_out/e_autotune_miss.txt
# unprofiled and profiled bench of the SAME command on the SAME box (rocprof-vs-wall reconciliation)
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/e_bench_c2.json.log 2>/dev/null; tail -1 gpurun_out/e_bench_c2.json.log | cut -c1-400
for c in c2 c3 c4; do
  mkdir -p gpurun_out/prof_$c
  ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$c -o bench -- python $REPO/bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/e_prof_$c.log 2>&1 )
  echo "rocprof $c rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/e_prof_$c.log | cut -c1-300
  find gpurun_out/prof_$c -name "*kernel_trace.csv" -delete
done
for c in c3 c4; do timeout 600 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/e_bench_$c.json.log 2>/dev/null; echo "$c rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/e_bench_$c.json.log | cut -c1-300; done
timeout 900 python bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/e_bench_c5.json.log 2>gpurun_out/e_bench_c5.err; echo "c5 rc=$? t=$(( $(date +%s) - T0 ))s"; tail -1 gpurun_out/e_bench_c5.json.log | cut -c1-400
mkdir -p gpurun_out/prof_c5
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_c5 -o bench -- python $REPO/bench.py --config c5 --batch 1 --sampler-steps 4 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $REPO/gpurun_out/e_prof_c5.log 2>&1 )
echo "rocprof c5 rc=$? t=$(( $(date +%s) - T0 ))s"
find gpurun_out/prof_c5 -name "*kernel_trace.csv" -delete
