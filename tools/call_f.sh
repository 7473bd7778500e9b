#!/bin/bash
# call F: the other BASELINE configs on one MI355X with the round-2 engine (shared CFG prefix): c3, c4, c5 (+ c5 with the tiled VAE)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 300 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/f_c3.log 2>&1; echo "c3 rc=$? t=$(( $(date +%s) - T0 ))s $(tail -1 gpurun_out/f_c3.log | cut -c1-200)"
timeout 400 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/f_c4.log 2>&1; echo "c4 rc=$? t=$(( $(date +%s) - T0 ))s $(tail -1 gpurun_out/f_c4.log | cut -c1-200)"
timeout 600 python bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/f_c5.log 2>&1; echo "c5 rc=$? t=$(( $(date +%s) - T0 ))s $(tail -1 gpurun_out/f_c5.log | cut -c1-200)"
timeout 600 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --vae-tiled on > gpurun_out/f_c4_vaetiled.log 2>&1; echo "c4 vae-tiled rc=$? t=$(( $(date +%s) - T0 ))s $(tail -1 gpurun_out/f_c4_vaetiled.log | cut -c1-200)"
