#!/bin/bash
# Socket power / clocks while the MFMA-only loop of tools/mfma_clock.hip runs for several seconds on zero and on random f16
# operands: is the clock drop on random operands an average-power cap (power pinned at the cap, clock settling
# over seconds) or something faster?  rocm-smi is sampled beside `mfma_clock sustain <fill> 5`.
# Usage (GPU box, repo root): bash tools/power_probe.sh [out]   -> <out> (text)
OUT=${1:-gpurun_out/power_probe.txt}
mkdir -p "$(dirname "$OUT")" gpurun_out
BIN=gpurun_out/mfma_clock
hipcc --offload-arch=gfx950 -O3 -w tools/mfma_clock.hip -o "$BIN" || exit 1
{
  echo "## idle: rocm-smi --showmaxpower --showpower --showclocks"
  rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -E "GPU\[0\]" | head -20
} > "$OUT"
for FILL in 0 3; do
  echo "## fill $FILL (0 = zeros, 3 = normal(0,1)): rocm-smi samples (power | sclk) beside mfma_clock sustain $FILL 5" >> "$OUT"
  "$BIN" sustain $FILL 5 > gpurun_out/mfma_sustain_$FILL.txt 2>&1 &
  PID=$!
  while kill -0 $PID 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "GPU\[0\].*(Power|sclk)" | sed 's/^.*GPU\[0\][ \t:]*//' | tr '\n' '|' >> "$OUT"
    echo >> "$OUT"
  done
  wait $PID
  cat gpurun_out/mfma_sustain_$FILL.txt >> "$OUT"
done
