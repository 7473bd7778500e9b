#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c11
mkdir -p $O
T0=$(date +%s)
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_multigpu_gpu.py -q -p no:cacheprovider -x > $O/pipe.log 2>&1
echo "pipeline+multigpu rc=$? $(tail -1 $O/pipe.log) t=$(( $(date +%s) - T0 ))s"; grep -E "^FAILED|^ERROR|Error|self-check" $O/pipe.log | head -8
