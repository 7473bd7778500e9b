"""Per-shape tile selection for the implicit-GEMM kernels.

`dbir_gemm` (C side) has a built-in heuristic (tile id 0).  On top of it this module holds a measured table
`tuning_gfx950.json` — problem key -> tile id — produced ON the MI355X by `tools/autotune.py`, which times every
eligible tile variant on the exact launches of the pipeline (same strides, epilogues, alignment) and validates each
variant's output against the default kernel before accepting it.  Lookups happen in `ops._gemm_launch`: exact key
first; otherwise the entry of the same problem class (mode, N, K, epilogue, stride ...) whose M is nearest within a
factor of two (the table is measured at the benchmark's batch — 16 samples per evaluation; other batches, e.g. config
C3's 8 or the tiled scheduler's 32-sample chunks, land on the neighbouring M bucket instead of the C heuristic; a
variant that cannot run the neighbouring shape is refused by `dbir_gemm` and the launch falls back to the heuristic).
A missing file or class falls back to the C heuristic.  `DBIR_TUNING=0` disables the table, `DBIR_TUNING_FILE`
overrides its path.
"""
import json
import os
from typing import Dict, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "tuning_gfx950.json")
_table: Optional[Dict[str, int]] = None
_classes: Dict[str, list] = {}      # key without M -> [(M, tile)]


def key_of(d) -> str:
    """Problem key of a GemmDesc: everything that changes which tile wins."""
    return (f"{d.mode}:{d.M}:{d.N}:{d.K}:a{d.act}:s{d.stride}:u{d.upsample}:z{max(d.batch, 1)}:"
            f"r{1 if d.R else 0}:v{1 if d.rowvec else 0}{':T' if d.store_mode else ''}{':f' if d.out_f32 else ''}")


def load(path: Optional[str] = None) -> Dict[str, int]:
    global _table
    if os.environ.get("DBIR_TUNING", "1") == "0":
        _table = {}
        return _table
    path = path or os.environ.get("DBIR_TUNING_FILE") or DEFAULT_PATH
    tab: Dict[str, int] = {}
    if os.path.exists(path):
        with open(path) as f:
            raw = json.load(f)
        for k, v in raw.get("tiles", raw).items():
            tab[k] = int(v["tile"] if isinstance(v, dict) else v)
    _table = tab
    _ensure_classes(tab)
    return tab


_classes_of = None  # the table object `_classes` was built from (tests / tools may install their own `_table`)


def _ensure_classes(tab) -> None:
    global _classes_of
    if _classes_of is tab:
        return
    _classes.clear()
    for k, t in tab.items():
        parts = k.split(":")
        _classes.setdefault(":".join(parts[:1] + parts[2:]), []).append((int(parts[1]), t))
    _classes_of = tab


def lookup_exact(d) -> Optional[int]:
    """The shipped table's entry for exactly this problem key, or None."""
    tab = _table if _table is not None else load()
    return tab.get(key_of(d))


def lookup(d) -> int:
    tab = _table if _table is not None else load()
    key = key_of(d)
    hit = tab.get(key)
    if hit is not None:
        return hit
    _ensure_classes(tab)
    parts = key.split(":")
    best, best_r = 0, 2.0 + 1e-9
    for m, t in _classes.get(":".join(parts[:1] + parts[2:]), ()):
        r = max(m, d.M) / max(min(m, d.M), 1)
        if r <= best_r and t:
            best, best_r = t, r
    # a borrowed entry keeps its tile shape but NOT its split-K factor: the slice count was measured for another M (the
    # workspace it implies scales with M, and captured HIP graphs hold pointers into the workspace); dbir_gemm still
    # refuses tiles the shape cannot run and the launch falls back to the C heuristic
    return best % 100
