cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/j_test_all.log 2>&1
echo "TEST all rc=$? : $(tail -1 gpurun_out/j_test_all.log)  t=$(( $(date +%s) - T0 ))s"; grep -E "^FAILED|^ERROR" gpurun_out/j_test_all.log | head -20
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/j_c2.log 2>&1; echo "c2 rc=$? $(tail -1 gpurun_out/j_c2.log | cut -c1-110)"
timeout 300 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/j_c3.log 2>&1; echo "c3 rc=$? $(tail -1 gpurun_out/j_c3.log | cut -c1-110)"
timeout 400 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/j_c4.log 2>&1; echo "c4 rc=$? $(tail -1 gpurun_out/j_c4.log | cut -c1-110)"
timeout 300 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/j_b1.log 2>&1; echo "b1 rc=$? $(tail -1 gpurun_out/j_b1.log | cut -c1-110) t=$(( $(date +%s) - T0 ))s"
