#!/bin/sh
# HBM traffic of the GEMM kernel family from rocprofv3 PMC counters, per the recipe of MI355X_MICROARCH.md (HBM /
# rocprofv3 PMC slots): FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) do not fit one pass -> two separate --pmc passes
# with --kernel-trace only; units are KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
# stream -> doubled in tools/pmc_summarize.py.  Workload: ONE batched network evaluation (ControlNet + UNet, 16 samples).
# Usage (on the GPU box, from the repo root): sh tools/pmc_traffic.sh [outdir] [batch]   -> <outdir>/pmc_traffic.json
# batch = images per evaluation half (8: the c2 evaluation of 16 samples; 4: c3; 16: the tiled scheduler's 32-sample chunks)
set -e
OUT=${1:-gpurun_out/pmc}
BATCH=${2:-8}
REPO=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
cd /tmp
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "gemm|attn|gn_|ln_kernel|ln2_kernel|splitk|xf_|xf2_" --output-format csv -d "$OUT/$C" -o pmc -- \
    python "$REPO/tools/profile_eval.py" --pmc-mode --pair --batch $BATCH > "$OUT/$C.log" 2>&1 || { tail -5 "$OUT/$C.log"; exit 1; }
done
cd "$REPO"
python tools/pmc_summarize.py "$OUT" "$OUT/pmc_traffic.json"
