"""ctypes binding of libdbir_hip.so (the C ABI declared in include/dbir.h).

The product path has NO fallback: if the shared library is missing or a call fails, we raise.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DBIR_HIP_LIB", os.path.join(_HERE, "libdbir_hip.so"))

F16, BF16 = 0, 1
ABI_VERSION = 5   # include/dbir.h dbir_abi_version(): checked at load time (a stale libdbir_hip.so must not run silently)
ACT_NONE, ACT_SILU, ACT_GELU, ACT_LRELU, ACT_GEGLU = 0, 1, 2, 3, 4
MODE_LINEAR, MODE_CONV3X3 = 0, 1


class GemmDesc(Structure):
    """Mirror of `dbir_gemm_desc` (include/dbir.h) — field order and types must match exactly."""
    _fields_ = [
        ("mode", c_int), ("dtype", c_int),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("A", c_void_p), ("lda", c_longlong), ("strideA_z", c_longlong),
        ("W", c_void_p), ("Wrows", c_int), ("Kpad", c_int), ("strideW_z", c_longlong),
        ("B", c_int), ("Hi", c_int), ("Wi", c_int), ("Cin", c_int), ("Ho", c_int), ("Wo", c_int),
        ("stride", c_int), ("pad", c_int), ("upsample", c_int),
        ("bias", c_void_p),
        ("rowvec", c_void_p), ("rowvec_ld", c_int), ("rows_per_batch", c_int),
        ("act", c_int), ("act_param", c_float), ("out_scale", c_float),
        ("R", c_void_p), ("ldr", c_longlong), ("strideR_z", c_longlong),
        ("C", c_void_p), ("ldc", c_longlong), ("strideC_z", c_longlong),
        ("out_f32", c_int), ("store_mode", c_int), ("trans_L", c_int),
        ("trans_ld", c_longlong), ("trans_bstride", c_longlong),
        ("batch", c_int), ("tile", c_int),
        ("splitk", c_int), ("ws", c_void_p), ("ws_bytes", c_longlong),
        ("stats", c_void_p), ("stats_rows", c_int),
    ]


_I, _LL, _F, _P = c_int, c_longlong, c_float, c_void_p

# name -> argtypes (restype is int unless noted).  Kept in one table so tests can check that the library
# exports every symbol the header declares.
SIGNATURES = {
    "dbir_gemm": [POINTER(GemmDesc), _P],
    "dbir_attention": [_I, _P, _LL, _LL, _P, _LL, _LL, _P, _LL, _LL, _P, _LL, _LL, _I, _I, _I, _I, _F, _P],
    "dbir_window_attention": [_I, _P, _LL, _P, _LL, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "dbir_groupnorm_nchunk": [_I, _I],
    "dbir_groupnorm": [_I, _P, _LL, _P, _LL, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P],
    "dbir_groupnorm_stats": [_I, _P, _LL, _I, _I, _I, _I, _P, _P, _P],
    "dbir_groupnorm_apply": [_I, _P, _LL, _P, _LL, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "dbir_groupnorm_from_partials": [_P, _I, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    "dbir_groupnorm_apply_partials": [_I, _P, _LL, _P, _LL, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "dbir_groupnorm_affine": [_I, _P, _LL, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    "dbir_layernorm": [_I, _P, _LL, _P, _LL, _P, _P, _I, _I, _I, _F, _P],
    "dbir_xf_tile_bytes": [], "dbir_xf_head_tiles": [], "dbir_xf_tail_tiles": [], "dbir_xf_geometry": [_I, _P, _P, _P],
    "dbir_xf2_geometry": [_I, _P, _P, _P, _P],
    "dbir_xf_head": [_I, _P, _LL, _P, _P, _LL, _P, _LL, _P, _LL, _LL, _I, _I, _I, _P, _LL, _P, _P],
    "dbir_xf_tail": [_I, _P, _LL, _P, _LL, _P, _LL, _P, _LL, _I, _I, _I, _I, _P, _LL, _P, _P, _P, _I, _F, _I, _P],
    "dbir_softmax_rows": [_I, _P, _LL, _LL, _I, _P],
    "dbir_clip_embed": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dbir_add_layernorm_f32": [_I, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _F, _P],
    "dbir_causal_attention": [_I, _P, _LL, _P, _LL, _I, _I, _I, _F, _P],
    "dbir_add_scaled": [_I, _P, _LL, _P, _LL, _F, _P, _LL, _LL, _I, _P],
    "dbir_block2x2": [_P, _LL, _P, _LL, _I, _I, _I, _I, _I, _P],
    "dbir_nchw_to_nhwc": [_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _F, _F, _P],
    "dbir_nhwc_to_nchw": [_I, _P, _I, _LL, _P, _I, _I, _I, _I, _F, _P, _P],
    "dbir_pixel_unshuffle": [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _F, _P],
    "dbir_timestep_embedding": [_I, _P, _P, _I, _I, _F, _P],
    "dbir_lincomb4": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _LL, _P],
    "dbir_spaced_step": [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I, _LL, _P],
    "dbir_tile_gather": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dbir_tile_accumulate": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dbir_tile_accumulate_partial": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dbir_tile_normalize": [_P, _P, _P, _LL, _LL, _P],
    "dbir_u8_to_f32_nchw": [_P, _P, _I, _I, _I, _P],
    "dbir_wavelet_blur": [_P, _P, _I, _I, _I, _I, _P],
    "dbir_colorfix": [_P, _P, _P, _P, _LL, _P],
    "dbir_f32_nchw_to_u8_nhwc": [_P, _P, _I, _I, _I, _P],
    "dbir_copy_rows": [_P, _LL, _P, _LL, _LL, _I, _P],
    "dbir_plan_fn_index": [c_char_p],
    "dbir_plan_create": [_P, _P, _I, _P, _LL, _I, _I],
    "dbir_plan_bind": [_P, _I, _P, _LL],
    "dbir_plan_run": [_P, _P],
    "dbir_plan_num_ops": [_P],
    "dbir_plan_destroy": [_P],
    "dbir_cldm_forward": [_P, _P, _P, _P, _P, _P],
    "dbir_abi_version": [],
    "dbir_set_option": [_I, _I],
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP kernel library; raises NativeError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or sh diffbir_amd/csrc/build.sh). There is no CPU / PyTorch fallback.")
        try:
            l = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 missing
            raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        l.dbir_last_error.restype = c_char_p
        l.dbir_last_error.argtypes = []
        got = int(l.dbir_abi_version())
        if got != ABI_VERSION:
            raise NativeError(f"{LIB_PATH} has ABI version {got}, this package needs {ABI_VERSION}: rebuild it "
                              "(sh diffbir_amd/csrc/build.sh)")
        _lib = l
        v = os.environ.get("DBIR_ATTN_VARIANT")  # A/B switch (include/dbir.h DBIR_OPT_ATTN_VARIANT)
        if v:
            check(l.dbir_set_option(1, int(v)), "dbir_set_option")
        v = os.environ.get("DBIR_XF_VARIANT")    # A/B switch (DBIR_OPT_XF_VARIANT)
        if v:
            check(l.dbir_set_option(2, int(v)), "dbir_set_option")
    return _lib


class _CountingLib:
    """Proxy over the loaded library that counts C-ABI calls (bench.py `launch_path.c_abi_calls_per_eval`)."""

    def __init__(self, l):
        object.__setattr__(self, "_l", l)
        object.__setattr__(self, "n", 0)

    def __getattr__(self, name):
        fn = getattr(self._l, name)
        if not callable(fn):
            return fn

        def counted(*a):
            object.__setattr__(self, "n", self.n + 1)
            return fn(*a)

        object.__setattr__(self, name, counted)
        return counted


def count_calls(on: bool) -> int:
    """on=True: start counting every call into the library; on=False: stop and return the count since the start."""
    global _lib
    l = lib()
    if on:
        if not isinstance(l, _CountingLib):
            _lib = _CountingLib(l)
        return 0
    if isinstance(l, _CountingLib):
        _lib = l._l
        return l.n
    return 0


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().dbir_last_error().decode(errors="replace")
        raise NativeError(f"{what} failed (status {status}): {msg}")
