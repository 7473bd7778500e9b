"""First-use tile autotuning for problem keys the shipped table does not hold.

`tuning_gfx950.json` has exact keys for the benchmarked configurations; an unlisted image size / batch lands on the
nearest-M entry of its problem class, which was measured to cost 20 - 40 % end to end.  With this module a table miss is
tuned ON the device the first time the launch is issued: the default tile and the tiles that ever win for the key's
problem class (mode, N, K, epilogue ...: a dozen candidates instead of ~90) are run on the launch's real operands,
validated against the default kernel's output, timed (min of 4 HIP-event timings on the launch stream), and the winner
is kept in `~/.cache/diffbir_amd/tuning_<device>.json` for later processes.  One key costs ~10 - 30 ms once; a new image
shape has ~100 keys.

The cache file carries the library's ABI version + candidate tile set (a mismatch discards it), keys include the element
type, and an entry `dbir_gemm` rejects is evicted.  Timing-based choices are not bit-reproducible across processes / ranks
(two tiles sum in different orders): set DBIR_AUTOTUNE=0 where bit-identical outputs across runs matter (the tests do).

Off: DBIR_AUTOTUNE=0.  Never runs while a HIP graph is being captured, for in-place residual launches (re-running would
accumulate), or for launches too small to matter (< 0.5 GFLOP).  Correctness never depends on it: every candidate's
output is compared with the default kernel's, and `dbir_gemm` refuses tiles a shape cannot run.
"""
import atexit
import ctypes
import json
import os
from typing import Dict, List, Optional

import torch

from . import native, tuning

ENABLED = os.environ.get("DBIR_AUTOTUNE", "1") != "0" and os.environ.get("DBIR_TUNING", "1") != "0"
MIN_FLOPS = 0.5e9
GENERIC = [5, 10, 14, 15, 25, 30, 34, 36, 37, 44, 45, 50, 52, 70, 71, 72, 73, 80, 95, 280, 480]
ALWAYS = [14, 15, 25, 36, 37, 50, 52, 70, 71, 72, 73, 80, 95, 280, 480]   # 95: register-streaming 64 x 80 (wins at M = 1024, K = 1280)
# (the producer / consumer tiles 90 - 92 win 5 - 15 % on K >= 1280 linears timed alone and nothing / -2 % in the two-stream
# evaluation, profiles/r3_pc_tiles_ab.txt: they are not offered to the timing-based choice)   # added behind a class's own winners (dbir_gemm refuses misfits)
_cache: Optional[Dict[str, int]] = None
_cache_path: Optional[str] = None
_dirty = False
_class_cands: Dict[str, List[int]] = {}
_class_src = None
stats = dict(tuned=0, hits=0, us_spent=0.0)


def _signature() -> str:
    """What a cached winner depends on besides the problem key: the library's ABI version and the candidate tile set
    (tiles were added and retired between builds; a stale entry made every such launch fail and retry)."""
    try:
        abi = int(native.lib().dbir_abi_version())
    except Exception:  # noqa: BLE001 - no library (CPU-only process): the cache is never consulted for a launch anyway
        abi = -1
    return f"abi{abi}:" + ",".join(str(t) for t in sorted(set(GENERIC + ALWAYS)))


def cache_key(d) -> str:
    """tuning.key_of + the element type (f16 and bf16 winners differ: the matrix pipe clocks differently)."""
    return f"{tuning.key_of(d)}:d{d.dtype}"


def evict(key: str) -> None:
    """Drop an entry `dbir_gemm` rejected (ops._gemm_launch): the next launch of the key is tuned again."""
    global _dirty
    if _cache is not None and _cache.pop(key, None) is not None:
        _dirty = True


def _path() -> str:
    name = "cpu"
    if torch.cuda.is_available():
        name = torch.cuda.get_device_name(0).replace(" ", "_").replace("/", "_")
    base = os.environ.get("DBIR_AUTOTUNE_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "diffbir_amd")
    return os.path.join(base, f"tuning_{name}.json")


def _load() -> Dict[str, int]:
    global _cache, _cache_path
    if _cache is None:
        _cache_path = _path()
        _cache = {}
        try:
            with open(_cache_path) as f:
                raw = json.load(f)
            if raw.get("signature") == _signature():   # otherwise: another library / tile set wrote it -> start over
                _cache = {k: int(v) for k, v in raw.get("tiles", {}).items()}
        except (OSError, ValueError):
            pass
        atexit.register(save)
    return _cache


def save() -> None:
    global _dirty
    if not _dirty or _cache is None:
        return
    try:
        os.makedirs(os.path.dirname(_cache_path), exist_ok=True)
        tmp = _cache_path + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(dict(signature=_signature(), tiles=_cache), f, indent=0, sort_keys=True)
        os.replace(tmp, _cache_path)
        _dirty = False
    except OSError:
        pass


def class_of(key: str) -> str:
    parts = key.split(":")
    return ":".join(parts[:1] + parts[2:])


def candidates(key: str) -> List[int]:
    """Tile codes worth timing for `key`: every winner of its problem class in the shipped table (any M), else a generic
    shortlist; split-K codes only for small-M deep-K problems (as tools/autotune.py)."""
    global _class_src
    tab = tuning._table if tuning._table is not None else tuning.load()
    if _class_src is not tab:
        _class_cands.clear()
        for k, t in tab.items():
            if t:
                c = _class_cands.setdefault(class_of(k), [])
                if t not in c:
                    c.append(t)
        _class_src = tab
    c = list(_class_cands.get(class_of(key), ()))
    c += [t for t in (ALWAYS if c else GENERIC) if t not in c]
    return c[:24]


def lookup(key: str) -> Optional[int]:
    if not ENABLED:
        return None
    hit = _load().get(key)
    if hit is not None:
        stats["hits"] += 1
    return hit


def tune(d, out: torch.Tensor, apply_tile_code) -> Optional[int]:
    """Time the candidates of this launch on its real operands; returns the winning code (0 = C heuristic) or None when
    the launch must not be re-run.  The caller launches once more with the returned code (a valid output is left behind)."""
    global _dirty
    from . import plan
    if not ENABLED or torch.cuda.is_current_stream_capturing() or plan.recording():
        return None
    if d.R and d.R == d.C:
        return None
    if 2.0 * d.M * d.N * d.K * max(d.batch, 1) < MIN_FLOPS:
        return None
    key = tuning.key_of(d)
    ckey = cache_key(d)
    lib, st = native.lib(), torch.cuda.current_stream().cuda_stream

    def timed(iters=4):
        best = float("inf")
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if lib.dbir_gemm(ctypes.byref(d), st) != 0:
                return None
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        return best

    apply_tile_code(d, 0, out.device)
    if lib.dbir_gemm(ctypes.byref(d), st) != 0:
        return None
    torch.cuda.synchronize()
    ref = out.float().clone()
    scale = ref.abs().max().item() + 1e-12
    us = {0: timed()}
    if us[0] is None:
        return None
    small_deep = d.M * d.N <= 160 * 256 * 256 and d.K >= 1280
    for c in candidates(key):
        if c >= 100 and not small_deep:
            continue
        apply_tile_code(d, c, out.device)
        out.zero_()
        if lib.dbir_gemm(ctypes.byref(d), st) != 0:
            continue
        torch.cuda.synchronize()
        if not ((out.float() - ref).abs().max().item() / scale <= 2e-2):
            continue
        t = timed()
        if t is not None:
            us[c] = t
    best = min(us, key=lambda k: us[k])
    if us[best] > 0.97 * us[0]:
        best = 0
    _load()[ckey] = int(best)
    _dirty = True
    stats["tuned"] += 1
    stats["us_spent"] += sum(v for v in us.values() if v) * 5
    return int(best)
