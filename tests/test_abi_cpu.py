"""CPU-only checks of the C-ABI boundary (no compute calls): the shared library builds for gfx950, loads, and exports
every function `include/dbir.h` declares; the ctypes mirror of `dbir_gemm_desc` matches the C layout; the product path
fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from diffbir_amd import native
    if not os.path.exists(native.LIB_PATH):
        subprocess.run(["sh", os.path.join(ROOT, "diffbir_amd", "csrc", "build.sh")], check=True)
    return native.LIB_PATH


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "dbir.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dbir_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib_path):
    from diffbir_amd import native
    lib = ctypes.CDLL(lib_path)
    declared = _declared_functions()
    assert "dbir_gemm" in declared and "dbir_attention" in declared and len(declared) >= 20
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, f"libdbir_hip.so does not export: {missing}"
    unbound = [n for n in declared if n not in native.SIGNATURES and n != "dbir_last_error"]
    assert not unbound, f"diffbir_amd.native.SIGNATURES lacks: {unbound}"
    assert lib.dbir_abi_version() >= 3


def test_gemm_desc_layout_matches_header(tmp_path):
    from diffbir_amd.native import GemmDesc
    fields = ["mode", "tile", "splitk", "ws", "ws_bytes", "C", "trans_bstride", "stats", "stats_rows"]
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu", sizeof(dbir_gemm_desc));%s return 0;}'
                    % (os.path.join(ROOT, "include", "dbir.h"),
                       "".join('printf(" %%zu", offsetof(dbir_gemm_desc, %s));' % f for f in fields)))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", str(prog), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(GemmDesc)
    assert out[1:] == [getattr(GemmDesc, f).offset for f in fields]


def test_product_path_has_no_cpu_fallback(lib_path):
    from diffbir_amd import native, ops
    with pytest.raises(native.NativeError):
        ops.layernorm(torch.zeros(4, 64, dtype=torch.float16), torch.ones(64), torch.zeros(64))
    # nothing under diffbir_amd/ may import the oracle or the test double
    for dirpath, _, files in os.walk(os.path.join(ROOT, "diffbir_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", text, flags=re.M), os.path.join(dirpath, f)


def test_tuning_table_is_wellformed():
    import json
    from diffbir_amd import tuning
    tab = tuning.load(tuning.DEFAULT_PATH)
    assert tab, "diffbir_amd/tuning_gfx950.json missing or empty"
    raw = json.load(open(tuning.DEFAULT_PATH))["tiles"]
    for k, v in raw.items():
        tile, sk = int(v["tile"]) % 100, int(v["tile"]) // 100
        assert (0 <= tile <= 73 or 80 <= tile <= 88 or tile == 95) and 0 <= sk <= 16, (k, v["tile"])   # tiles that exist in gemm*.hip (95: gemm_rs.hip)


def test_first_use_autotune_candidates_and_cache(tmp_path, monkeypatch):
    """Table misses are tuned on first use (diffbir_amd/autotune.py): candidate list = the problem class's winners in the
    shipped table + a short generic list; results persist in a per-device cache file; split-K codes are never borrowed by
    the nearest-M fallback."""
    import json
    from diffbir_amd import autotune, tuning
    tab = tuning.load(tuning.DEFAULT_PATH)
    key = next(k for k, v in tab.items() if v and v < 100)
    cls_winners = {v for k, v in tab.items() if autotune.class_of(k) == autotune.class_of(key) and v}
    cands = autotune.candidates(key)
    assert 0 < len(cands) <= 24 and len(set(cands)) == len(cands)
    assert set(cands[: len(cls_winners)]) <= cls_winners | set(autotune.ALWAYS)
    assert autotune.candidates("0:12345:7:9:a0:s0:u0:z1:r0:v0")[:3] == autotune.GENERIC[:3]
    monkeypatch.setenv("DBIR_AUTOTUNE_CACHE", str(tmp_path))
    monkeypatch.setattr(autotune, "_cache", None)
    monkeypatch.setattr(autotune, "ENABLED", True)
    assert autotune.lookup("some:key") is None
    autotune._load()["some:key"] = 37
    monkeypatch.setattr(autotune, "_dirty", True)
    autotune.save()
    files = list(tmp_path.iterdir())
    assert len(files) == 1 and json.load(open(files[0]))["tiles"] == {"some:key": 37}
    monkeypatch.setattr(autotune, "_cache", None)
    assert autotune.lookup("some:key") == 37

    class D:   # a key that is not in the table: the nearest-M fallback keeps the tile but drops the split-K factor
        mode, N, K, act, stride, upsample, batch, R, rowvec, store_mode, out_f32 = 1, 1280, 11520, 0, 1, 0, 1, 1, None, 0, 0
    d = D()
    d.M = 4096 + 8
    assert tuning.lookup_exact(d) is None and 0 <= tuning.lookup(d) < 100
