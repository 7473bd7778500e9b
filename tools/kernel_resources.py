#!/usr/bin/env python3
"""Register / LDS / scratch budget of every kernel in diffbir_amd/csrc, from the compiler's own resource remarks
(`-Rpass-analysis=kernel-resource-usage`, device-only compile with the flags of build.sh — no GPU needed).
Usage: python tools/kernel_resources.py [file.hip ...] > profiles/rN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffbir_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result --cuda-device-only -Rpass-analysis=kernel-resource-usage"
FIELDS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill", "Occupancy [waves/SIMD]",
          "LDS Size [bytes/block]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return [re.sub(r"\(anonymous namespace\)::", "", l).split("(")[0].replace("void ", "") for l in out.splitlines()]


def main():
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    print("# kernel resource usage, gfx950, flags of diffbir_amd/csrc/build.sh (tools/kernel_resources.py)")
    print("# LDS = static bytes only (the GEMM / fused kernels size their LDS at launch: dynamic shared memory)")
    print(f"# {'kernel':<86} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch B':>9} {'v-spill':>7} {'s-spill':>7} {'waves/SIMD':>10} {'LDS B':>7}")
    for f in files:
        r = subprocess.run(f"/opt/rocm/bin/hipcc {FLAGS} -c {f} -o /dev/null", shell=True, cwd=CSRC, capture_output=True, text=True)
        rows, cur = [], None
        for line in r.stderr.splitlines():
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = dict(name=m.group(1))
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1)] = m.group(2)
        if not rows:
            continue
        print(f"## {f}")
        for row, name in zip(rows, demangle([r_["name"] for r_ in rows])):
            v = [row.get(k, "?") for k in FIELDS]
            print(f"  {name[:86]:<86} {v[0]:>5} {v[1]:>5} {v[2]:>5} {v[3]:>9} {v[4]:>7} {v[5]:>7} {v[6]:>10} {v[7]:>7}")


if __name__ == "__main__":
    main()
