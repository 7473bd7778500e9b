"""Per-shape tile selection for the implicit-GEMM kernels.

`dbir_gemm` (C side) has a built-in heuristic (tile id 0).  On top of it this module holds a measured table
`tuning_gfx950.json` — problem key -> tile id — produced ON the MI355X by `tools/autotune.py`, which times every
eligible tile variant on the exact launches of the pipeline (same strides, epilogues, alignment) and validates each
variant's output against the default kernel before accepting it.  Lookups happen in `ops._gemm_launch`; a missing file
or key falls back to the C heuristic.  `DBIR_TUNING=0` disables the table, `DBIR_TUNING_FILE` overrides its path.
"""
import json
import os
from typing import Dict, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "tuning_gfx950.json")
_table: Optional[Dict[str, int]] = None


def key_of(d) -> str:
    """Problem key of a GemmDesc: everything that changes which tile wins."""
    return (f"{d.mode}:{d.M}:{d.N}:{d.K}:a{d.act}:s{d.stride}:u{d.upsample}:z{max(d.batch, 1)}:"
            f"r{1 if d.R else 0}:v{1 if d.rowvec else 0}{':T' if d.store_mode else ''}")


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
    return tab


def lookup(d) -> int:
    tab = _table if _table is not None else load()
    return tab.get(key_of(d), 0)
