#!/bin/sh
# Effective shader clock and matrix-pipe utilisation PER KERNEL of one batched network evaluation, from rocprofv3 PMC counters
# (one --pmc pass with --kernel-trace only): GRBM_GUI_ACTIVE / kernel duration = clock under that kernel's load,
# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) = share of matrix-pipe cycles in use.  A kernel that runs far below
# the clock of the bare MFMA loop (tools/mfma_clock.hip: 1.72 GHz on random f16 data) is power-limited: schedule changes return
# as clock, only fewer bytes / instructions per FLOP help.
# Usage (GPU box, repo root): sh tools/pmc_clock.sh [outdir] [batch]  ->  <outdir>/pmc_clock.txt
set -e
OUT=${1:-gpurun_out/pmc_clock}
BATCH=${2:-8}
REPO=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
cd /tmp
export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pass" -o pmc -- \
  python "$REPO/tools/profile_eval.py" --pmc-mode --pair --batch $BATCH > "$OUT/pass.log" 2>&1 || { tail -5 "$OUT/pass.log"; exit 1; }
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = {}
for fn in glob.glob(os.path.join(out, "pass", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
seen = set()
for fn in glob.glob(os.path.join(out, "pass", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        did = r["Dispatch_Id"]
        name = r["Kernel_Name"]
        rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if (did, name) not in seen:
            seen.add((did, name))
            cnt[name] += 1
            if did in dur:
                rows[name]["_ns"] += dur[did][1]
lines = []
for name, v in rows.items():
    n, ns, act = cnt[name], v.get("_ns", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    if n == 0 or ns <= 0 or act <= 0:
        continue
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3 (one value per XCC instance): per-XCD cycles = act / 8
    clk = act / 8.0 / ns * 1e3     # MHz
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act / 8.0 * 1024.0)
    lines.append((ns, f"{ns / n / 1e3:9.1f} us x {n:4d}  clock {clk:6.0f} MHz  MFMA busy {100 * mf:5.1f} %  {name[:110]}"))
with open(os.path.join(out, "pmc_clock.txt"), "w") as f:
    f.write("# per kernel of one evaluation (batch 16): avg duration x launches, GRBM_GUI_ACTIVE / duration, MFMA busy share\n")
    for _, ln in sorted(lines, reverse=True)[:45]:
        f.write(ln + "\n")
print(open(os.path.join(out, "pmc_clock.txt")).read())
PY
find "$OUT" -name "*kernel_trace.csv" -delete
