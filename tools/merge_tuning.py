#!/usr/bin/env python
"""Merge an incremental autotune result (tools/autotune.py --only ...) into diffbir_amd/tuning_gfx950.json: a key's tile
changes only when a candidate beats the incumbent RE-TIMED IN THE SAME PROCESS by more than 3 % (boxes differ in clock;
only within-run comparisons count).  Usage: python tools/merge_tuning.py gpurun_out/r2/tuning_x.json [--dry]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = os.path.join(ROOT, "diffbir_amd", "tuning_gfx950.json")
    cur = json.load(open(path))
    new = json.load(open(sys.argv[1]))
    dry = "--dry" in sys.argv
    # --exclude 90,91,92: never adopt these tiles (the producer / consumer tiles win timed alone and lose in the two-stream
    # evaluation, profiles/r3_pc_tiles_ab.txt: diffbir_amd/autotune.py does not offer them either)
    excl = set()
    if "--exclude" in sys.argv:
        excl = {int(x) % 100 for x in sys.argv[sys.argv.index("--exclude") + 1].split(",") if x}
    n = 0
    for k, v in sorted(new["tiles"].items()):
        us = {t: round(u, 1) for t, u in v["us"].items() if u and int(t) % 100 not in excl}
        if not us:
            continue
        ent = cur["tiles"].setdefault(k, dict(tile=0, us={}))
        prev = str(ent["tile"])
        if prev not in us:      # the table's tile was not re-timed in this run: nothing comparable (boxes differ)
            ent["us"].update({t: u for t, u in us.items() if t not in ent["us"]})
            continue
        inc = prev
        best = min(us, key=lambda t: us[t])
        if best != inc and us[best] < 0.97 * us[inc]:
            print(f"{k:50s} {prev:>5s} -> {best:>5s}   {us[inc]:8.1f} -> {us[best]:8.1f} us")
            ent["tile"] = int(best)
            n += 1
        ent["us"].update(us)
    print(n, "keys updated")
    if not dry:
        json.dump(cur, open(path, "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
