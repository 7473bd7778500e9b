"""Thin tensor-level wrappers over the C ABI (include/dbir.h).

PyTorch is plumbing only here: it owns device memory and the current HIP stream; every function below
validates shapes/strides, then enqueues hand-written gfx950 kernels from libdbir_hip.so on
`torch.cuda.current_stream()`.  There is no eager / CPU fallback: calling any op without the built library
or with non-GPU tensors raises.

Conventions
  * activations are 16-bit (`torch.float16` / `torch.bfloat16`) channels-last: 4-D `[B, H, W, C]` or 2-D `[M, C]`,
    unit stride in the last dim; the row stride (`ld`) may exceed C (views into wider buffers — this is how
    torch.cat of the reference (controlnet.py:41-43) is replaced by producers writing into one buffer).
  * weights are pre-packed once (`pack_*`) into the `[N_pad, K_pad]` K-contiguous layout the MFMA kernel reads.
"""
import ctypes
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import autotune, native, tuning
from .native import (ACT_GEGLU, ACT_GELU, ACT_LRELU, ACT_NONE, ACT_SILU, MODE_CONV3X3, MODE_LINEAR,  # noqa: F401
                     GemmDesc)

T = torch.Tensor


# ------------------------------------------------------------------------------------------------ helpers
def _dt(t: T) -> int:
    if t.dtype == torch.float16:
        return native.F16
    if t.dtype == torch.bfloat16:
        return native.BF16
    raise TypeError(f"expected a 16-bit tensor, got {t.dtype}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise native.NativeError("diffbir_amd ops need GPU tensors (no CPU fallback in the product path)")


def _ld(t: T) -> int:
    """Row stride (elements) of a channels-last activation whose leading dims are dense over that stride."""
    assert t.stride(-1) == 1, "last dim must be unit-stride"
    if t.dim() == 1:
        return t.shape[0]
    ld = t.stride(-2)
    exp = ld
    for d in range(t.dim() - 2, 0, -1):
        exp *= t.shape[d]
        assert t.stride(d - 1) == exp or t.shape[d - 1] == 1, f"leading dims not dense: {t.shape} {t.stride()}"
    return ld


def _rows(t: T) -> int:
    n = 1
    for s in t.shape[:-1]:
        n *= s
    return n


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------------------ packing
@dataclass
class PackedWeight:
    w: T                      # [Wrows, Kpad] 16-bit, K-contiguous
    bias: Optional[T]         # f32 [N] (packed column order) or None
    N: int                    # packed columns (== 2*N_out for GEGLU)
    K: int                    # logical K
    Kpad: int
    n_out: int                # columns actually written
    cin: int = 0              # conv3x3: (padded) input channels
    geglu: bool = False
    up4: Optional["PackedWeight"] = None   # conv3x3 behind a nearest-x2 upsample: the four parity-collapsed 2x2 weight sets


def _finish_pack(w2d: T, bias: Optional[T], dtype, device, n_out=None, cin=0, geglu=False) -> PackedWeight:
    N, K = w2d.shape
    Kpad = _rup(K, 64)
    Wrows = _rup(N, 128)
    buf = torch.zeros((Wrows, Kpad), dtype=dtype, device=device)
    buf[:N, :K] = w2d.to(device=device, dtype=dtype)
    b = None if bias is None else bias.to(device=device, dtype=torch.float32).contiguous()
    return PackedWeight(buf, b, N, K, Kpad, n_out if n_out is not None else N, cin, geglu)


def pack_linear(w: T, bias: Optional[T], dtype, device, k_pad_to: int = 8, n_pad_to: int = 1) -> PackedWeight:
    """nn.Linear / 1x1-conv weight [N, K(,1,1)] -> packed. K is zero-padded to a multiple of `k_pad_to`
    (activations with padded channel counts, e.g. SwinIR 180 -> 192) and N to `n_pad_to` (extra zero columns)."""
    w = w.reshape(w.shape[0], -1).float()
    N, K = w.shape
    Kp, Np = _rup(K, k_pad_to), _rup(N, n_pad_to)
    if Kp != K or Np != N:
        w2 = torch.zeros((Np, Kp), dtype=torch.float32, device=w.device)
        w2[:N, :K] = w
        w = w2
        if bias is not None:
            b2 = torch.zeros(Np, dtype=torch.float32, device=w.device)
            b2[:N] = bias.float()
            bias = b2
    return _finish_pack(w, bias, dtype, device)


def pack_geglu(w: T, bias: T, dtype, device) -> PackedWeight:
    """GEGLU projection [2*Nh, K] (first half values, second half gates; attention.py:24-26) -> rows interleaved in
    blocks of 32 (value block, gate block) so a wave's two 32-column MFMA tiles hold matching value/gate columns."""
    w = w.float()
    N2, K = w.shape
    nh = N2 // 2
    assert nh % 32 == 0, "GEGLU half width must be a multiple of 32"
    val, gate = w[:nh].reshape(nh // 32, 32, K), w[nh:].reshape(nh // 32, 32, K)
    wi = torch.stack([val, gate], dim=1).reshape(N2, K)
    bv, bg = bias.float()[:nh].reshape(nh // 32, 32), bias.float()[nh:].reshape(nh // 32, 32)
    bi = torch.stack([bv, bg], dim=1).reshape(N2)
    return _finish_pack(wi, bi, dtype, device, n_out=nh, geglu=True)


def pack_conv3x3(w: T, bias: Optional[T], dtype, device, cin_pad_to: int = 8, n_pad_to: int = 1,
                 up4: bool = False) -> PackedWeight:
    """Conv2d weight [N, Cin, 3, 3] -> [N, (ky, kx, c)] with Cin zero-padded to a multiple of `cin_pad_to`.
    up4: the convolution follows a nearest-x2 upsample (unet.py:76, vae.py:34): also pack the parity-collapsed form
    (`pack_conv3x3_up4`) when the channel counts allow it."""
    w = w.float()
    N, Cin = w.shape[:2]
    Cp, Np = _rup(Cin, cin_pad_to), _rup(N, n_pad_to)
    w2 = torch.zeros((Np, 3, 3, Cp), dtype=torch.float32, device=w.device)
    w2[:N, :, :, :Cin] = w.permute(0, 2, 3, 1)
    if bias is not None and Np != N:
        b2 = torch.zeros(Np, dtype=torch.float32, device=w.device)
        b2[:N] = bias.float()
        bias = b2
    pw = _finish_pack(w2.reshape(Np, 9 * Cp), bias, dtype, device, cin=Cp)
    if up4 and UP4 and Cp % 32 == 0 and Np == N and N % 8 == 0:
        pw.up4 = pack_conv3x3_up4(w2, pw.bias, dtype, device)
    return pw


# Nearest-x2 upsample followed by a 3x3 convolution (reference unet.py:51-79 `Upsample`, vae.py:24-36): output pixel
# (2i + a, 2j + b) only ever sees the 2x2 low-resolution neighbourhood (i + a - 1 .., j + b - 1 ..), and the 3x3 taps that
# land on the same low-resolution pixel can be summed ahead of time: row taps {ky} -> a = 0: [W0], [W1 + W2]; a = 1:
# [W0 + W1], [W2] (same for columns).  Four 2x2 convolutions (one per output parity) replace one 3x3 convolution on a 4x
# larger grid: 16 instead of 36 multiplies per low-resolution pixel and channel pair — 4 / 9 of the work, exact algebra
# (the only numerical difference: the f32 tap sums are rounded to 16 bit once, instead of each tap being rounded).
# DBIR_UP4=0 keeps the upsampled 3x3 gather (A/B).
UP4 = os.environ.get("DBIR_UP4", "1") != "0"
_UP4_TAPS = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}   # (parity, low-res tap) -> 3x3 taps


def up4_weights(w_khwc: T) -> T:
    """[N, 3, 3, C] f32 -> [4 (parity a * 2 + b), N, 2 (ty), 2 (tx), C] f32 tap sums."""
    N, _, _, C = w_khwc.shape
    w4 = torch.zeros((4, N, 2, 2, C), dtype=torch.float32, device=w_khwc.device)
    for a in range(2):
        for b in range(2):
            for ty in range(2):
                for tx in range(2):
                    for ky in _UP4_TAPS[(a, ty)]:
                        for kx in _UP4_TAPS[(b, tx)]:
                            w4[2 * a + b, :, ty, tx] += w_khwc[:, ky, kx]
    return w4


def pack_conv3x3_up4(w_khwc: T, bias_f32: Optional[T], dtype, device) -> PackedWeight:
    """[N, 3, 3, Cin] (Cin % 32 == 0) -> PackedWeight whose `w` is [4 * Wrows, 4 * Cin]: four [Wrows, (ty, tx, c)] matrices."""
    N, _, _, C = w_khwc.shape
    w4 = up4_weights(w_khwc.float())
    Wrows, K = _rup(N, 128), 4 * C
    buf = torch.zeros((4, Wrows, K), dtype=dtype, device=device)
    buf[:, :N] = w4.reshape(4, N, K).to(device=device, dtype=dtype)
    return PackedWeight(buf.reshape(4 * Wrows, K), bias_f32, N, K, K, N, C)


# ------------------------------------------------------------------------------------------------ GEMM family
_PROFILE = None  # list of (kind, algorithmic_flops, start_event, end_event) while profiling (bench.py roofline)


def start_profile():
    global _PROFILE
    _PROFILE = []
    return _PROFILE


def stop_profile():
    global _PROFILE
    rec, _PROFILE = _PROFILE, None
    return rec


class _Timed:
    """Brackets ONE kernel launch with HIP events on the current stream when profiling is on (else free)."""

    def __init__(self, kind: str, flops: float, tag: str = "", nbytes: float = 0.0):
        self.kind, self.flops, self.tag, self.nbytes = kind, flops, tag, nbytes

    def __enter__(self):
        if _PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if _PROFILE is not None:
            self.e1.record()
            _PROFILE.append((self.kind, self.flops, self.e0, self.e1, self.tag, self.nbytes))
        return False


@dataclass
class GnPartials:
    """Column statistics of a GEMM output emitted by its epilogue (dbir_gemm_desc.stats, ABI >= 4): buf f32 [M / rows, 2, N] —
    per tile of `rows` rows and column: [0] the SUM of the stored 16-bit values, [1] their M2 = the sum of squared deviations
    from that tile-column's own mean (NOT the sum of squares: merge tiles with Chan's parallel-variance formula, as
    csrc/norm.hip gn_from_partials / gn_apply_partials do).  Every tile lies inside one sample; the tiles of a sample are
    adjacent rows of `buf` (in any order).  Consumed by groupnorm(..., stats=...)."""
    buf: T
    rows: int
    N: int
    M: int


def _stats_begin(d: GemmDesc, want: bool, M: int, N: int, device):
    if not want:
        return None
    st = torch.empty(((M + 63) // 64) * 2 * N, dtype=torch.float32, device=device)
    d.stats = st.data_ptr()
    return st


def _stats_end(d: GemmDesc, st: Optional[T], M: int, N: int) -> Optional[GnPartials]:
    if st is None or d.stats_rows <= 0:
        return None
    rows = int(d.stats_rows)
    return GnPartials(st[: (M // rows) * 2 * N], rows, N, M)


_TUNER = None  # tools/autotune.py installs an object with .run(d, out) while measuring tile variants
_WS = {}       # device -> f32 split-K workspace (grown on demand, reused by every launch on that device's stream)
_WS_KEEP = []  # superseded workspaces: captured HIP graphs (model/cldm.py) hold raw pointers into them, so a regrown
               # workspace must never be returned to the allocator while the process lives (a few hundred MB at most:
               # sizes at least double)


def splitk_workspace(device, nbytes: int) -> T:
    """One workspace per (device, stream): launches on different streams may run concurrently."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _WS_KEEP.append(ws)
            nbytes = max(nbytes, 2 * ws.numel() * 4)
        ws = torch.empty(max(nbytes, 64 << 20) // 4, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def apply_tile_code(d: GemmDesc, code: int, device) -> None:
    """code = tile id + 100 * split-K factor (diffbir_amd/tuning.py)."""
    d.tile, sk = code % 100, code // 100
    if sk > 1:
        if d.tile == 80:   # in-launch reduce: whole 256 x 320 accumulator slabs per (tile, slice) + arrival counters
            mt = 4 * ((d.M // 4 + 255) // 256) if d.upsample == 2 else (d.M + 255) // 256   # (up4: row tiles per parity)
            tiles = mt * ((d.N + 319) // 320)
            nbytes = tiles * sk * 256 * 320 * 4 + tiles * 4 + 256
        else:
            nbytes = sk * max(d.batch, 1) * d.M * d.N * 4
        ws = splitk_workspace(device, nbytes)
        d.splitk, d.ws, d.ws_bytes = sk, ws.data_ptr(), ws.numel() * 4
    else:
        d.splitk, d.ws, d.ws_bytes = 0, None, 0


def _gemm_launch(d: GemmDesc, keep):
    out = keep[2]
    from_table = False
    if d.tile == 0 and d.upsample == 2:
        raise native.NativeError("conv3x3: the parity-collapsed upsample convolution needs an explicit tile-80 code")
    if d.tile == 0:
        if _TUNER is not None:
            code = _TUNER.run(d, out)
        else:
            code = tuning.lookup_exact(d)
            if code is None:   # not in the shipped table: this device's first-use autotune cache, tuned on a miss
                key = autotune.cache_key(d)
                code = autotune.lookup(key)
                if code is None:
                    code = autotune.tune(d, out, apply_tile_code)
                if code is None:
                    code = tuning.lookup(d)   # nearest-M entry of the problem class, then the C heuristic
        from_table = code != 0
        apply_tile_code(d, code, out.device)
    tag = ""
    if _PROFILE is not None:
        tag = (f"{'conv' if d.mode == MODE_CONV3X3 else 'lin'} M{d.M} N{d.N} K{d.K} z{max(d.batch, 1)} act{d.act}"
               f"{' s2' if d.stride == 2 else ''}{(' up4' if d.upsample == 2 else ' up') if d.upsample else ''}{' T' if d.store_mode else ''}"
               f"{' f32' if d.out_f32 else ''}{' res' if d.R else ''} t{d.tile}{'k%d' % d.splitk if d.splitk > 1 else ''}")
    # algorithmic HBM bytes: every operand once (conv input once, not once per tap), 16-bit
    z = max(d.batch, 1)
    a_elems = float(d.B) * d.Hi * d.Wi * d.Cin if d.mode == MODE_CONV3X3 else float(d.M) * d.K
    n_st = d.N // 2 if d.act == ACT_GEGLU else d.N
    nbytes = z * (2.0 * a_elems + 2.0 * d.N * d.K + (4.0 if d.out_f32 else 2.0) * d.M * n_st
                  + (2.0 * d.M * n_st if d.R else 0.0))
    with _Timed("gemm", 2.0 * d.M * (d.N) * d.K * z, tag, nbytes):
        rc = native.lib().dbir_gemm(ctypes.byref(d), _stream())
        if rc != 0 and from_table:  # a tuned variant whose alignment requirements this call does not meet
            autotune.evict(autotune.cache_key(d))
            apply_tile_code(d, 0, out.device)
            rc = native.lib().dbir_gemm(ctypes.byref(d), _stream())
        native.check(rc, "dbir_gemm")


def _fill_epilogue(d: GemmDesc, pw: Optional[PackedWeight], act, act_param, out_scale, residual, rowvec,
                   rows_per_batch, out: T, out_f32: bool):
    d.bias = pw.bias.data_ptr() if (pw is not None and pw.bias is not None) else None
    if rowvec is not None:
        d.rowvec, d.rowvec_ld, d.rows_per_batch = rowvec.data_ptr(), _ld(rowvec), rows_per_batch
    d.act, d.act_param, d.out_scale = act, act_param, out_scale
    if residual is not None:
        d.R, d.ldr = residual.data_ptr(), _ld(residual)
    d.C, d.ldc, d.out_f32 = out.data_ptr(), _ld(out), int(out_f32)
    d.batch = 1


def linear(x: T, pw: PackedWeight, out: Optional[T] = None, act: int = ACT_NONE, act_param: float = 0.0,
           out_scale: float = 1.0, residual: Optional[T] = None, rowvec: Optional[T] = None,
           rows_per_batch: int = 0, out_f32: bool = False, tile: int = 0, stats: bool = False):
    """out[..., :n_out] = epilogue(x[..., :K] @ W^T).  x: [..., K] 16-bit (ld >= K).
    stats=True -> returns (out, GnPartials | None): GroupNorm column sums of the output from the epilogue."""
    _gpu(x, out, residual, rowvec)
    M, K = _rows(x), x.shape[-1]
    assert K == pw.K, f"K mismatch: x has {K}, weight has {pw.K}"
    if pw.geglu:
        act = ACT_GEGLU
    if out is None:
        out = torch.empty(x.shape[:-1] + (pw.n_out,), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    assert out.shape[-1] == pw.n_out and _rows(out) == M
    d = GemmDesc()
    d.mode, d.dtype, d.M, d.N, d.K = MODE_LINEAR, _dt(x), M, pw.N, K
    d.A, d.lda = x.data_ptr(), _ld(x)
    d.W, d.Wrows, d.Kpad = pw.w.data_ptr(), pw.w.shape[0], pw.Kpad
    _fill_epilogue(d, pw, act, act_param, out_scale, residual, rowvec, rows_per_batch, out, out_f32)
    apply_tile_code(d, tile, x.device)  # tile id, or tile + 100 * split-K slices
    st = _stats_begin(d, stats, M, pw.N, x.device)
    _gemm_launch(d, (x, pw, out, residual, rowvec, st))
    return (out, _stats_end(d, st, M, pw.N)) if stats else out


def linear_t(x: T, pw: PackedWeight, L: int, out_t: T, tile: int = 0) -> T:
    """Transposed store: out_t[b, n, l] = (x @ W^T)[b*L + l, n] for x: [Bz*L, K]; out_t: [Bz, N, Lpad] 16-bit."""
    _gpu(x, out_t)
    M, K = _rows(x), x.shape[-1]
    assert K == pw.K and M % L == 0 and out_t.dim() == 3 and out_t.shape[1] == pw.n_out
    assert out_t.stride(2) == 1 and out_t.shape[0] == M // L and out_t.shape[2] >= L
    d = GemmDesc()
    d.mode, d.dtype, d.M, d.N, d.K = MODE_LINEAR, _dt(x), M, pw.N, K
    d.A, d.lda = x.data_ptr(), _ld(x)
    d.W, d.Wrows, d.Kpad = pw.w.data_ptr(), pw.w.shape[0], pw.Kpad
    d.bias = pw.bias.data_ptr() if pw.bias is not None else None
    d.act, d.out_scale = ACT_NONE, 1.0
    d.C, d.ldc = out_t.data_ptr(), 0
    d.store_mode, d.trans_L, d.trans_ld, d.trans_bstride = 1, L, out_t.stride(1), out_t.stride(0)
    d.batch = 1
    d.tile = tile
    _gemm_launch(d, (x, pw, out_t))
    return out_t


def conv3x3(x: T, pw: PackedWeight, stride: int = 1, pad: int = 1, upsample: bool = False,
            out: Optional[T] = None, act: int = ACT_NONE, act_param: float = 0.0, out_scale: float = 1.0,
            residual: Optional[T] = None, rowvec: Optional[T] = None, out_f32: bool = False,
            out_hw: Optional[Tuple[int, int]] = None, tile: int = 0, stats: bool = False):
    """3x3 convolution as implicit GEMM. x: [B, Hi, Wi, Cin] 16-bit DENSE (ld == Cin == pw.cin).
    `upsample`: nearest x2 fused into the gather. `out_hw` overrides the output extent (VAE asymmetric pad)."""
    _gpu(x, out, residual, rowvec)
    B, Hi, Wi, Cin = x.shape
    assert Cin == pw.cin and _ld(x) == Cin, f"conv3x3 needs dense NHWC input with C == {pw.cin}, got {x.shape} ld {_ld(x)}"
    Hv, Wv = (2 * Hi, 2 * Wi) if upsample else (Hi, Wi)
    if out_hw is None:
        Ho, Wo = (Hv + 2 * pad - 3) // stride + 1, (Wv + 2 * pad - 3) // stride + 1
    else:
        Ho, Wo = out_hw
    if out is None:
        out = torch.empty((B, Ho, Wo, pw.n_out), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    assert tuple(out.shape) == (B, Ho, Wo, pw.n_out)
    d = GemmDesc()
    d.mode, d.dtype, d.M, d.N, d.K = MODE_CONV3X3, _dt(x), B * Ho * Wo, pw.N, 9 * Cin
    d.A, d.lda = x.data_ptr(), Cin
    d.W, d.Wrows, d.Kpad = pw.w.data_ptr(), pw.w.shape[0], pw.Kpad
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo = B, Hi, Wi, Cin, Ho, Wo
    d.stride, d.pad, d.upsample = stride, pad, int(upsample)
    if upsample and _up4_ok(pw, x, out, stride, pad, out_hw, residual, rowvec, out_f32, tile):
        # parity-collapsed form (pack_conv3x3_up4): four 2x2 convolutions on the low-resolution grid, tile 80 only
        u = pw.up4
        d.K, d.W, d.Wrows, d.Kpad, d.upsample = u.K, u.w.data_ptr(), u.w.shape[0] // 4, u.Kpad, 2
        if tile == 0:   # one 256 x 320 tile per (parity, 256 low-resolution pixels, 320 columns); split K to fill the chip
            tiles = 4 * ((B * Hi * Wi + 255) // 256) * ((pw.N + 319) // 320)
            sk = 1 if tiles >= 192 else min(max(256 // tiles, 1), max(u.K // 32 // 16, 1))
            tile = 80 + (100 * sk if sk > 1 else 0)
        keep_w = u
    else:
        keep_w = pw
    _fill_epilogue(d, pw, act, act_param, out_scale, residual, rowvec, Ho * Wo, out, out_f32)
    apply_tile_code(d, tile, x.device)  # tile id, or tile + 100 * split-K slices
    st = _stats_begin(d, stats, B * Ho * Wo, pw.N, x.device)
    _gemm_launch(d, (x, keep_w, out, residual, rowvec, st))
    return (out, _stats_end(d, st, B * Ho * Wo, pw.N)) if stats else out


def _up4_ok(pw: PackedWeight, x: T, out: T, stride, pad, out_hw, residual, rowvec, out_f32, tile) -> bool:
    """Can this upsampled convolution run in the parity-collapsed form?  (mirrors dbir_gemm_8p_eligible)"""
    if pw.up4 is None or not UP4 or stride != 1 or pad != 1 or out_hw is not None or residual is not None \
            or rowvec is not None or out_f32 or (tile % 100) not in (0, 80):
        return False
    _, Hi, Wi, _ = x.shape
    if Hi & (Hi - 1) or Wi & (Wi - 1):
        return False
    ldc = _ld(out)
    return ldc % 8 == 0 and out.data_ptr() % 16 == 0 and (_rows(out) - 1) * ldc + pw.N < (1 << 30)


def bmm_nt(a: T, b: T, out: T, out_scale: float = 1.0) -> T:
    """Batched out[z] = a[z] @ b[z]^T * out_scale with a: [Z, M, K], b: [Z, N, K], out: [Z, M, N] (16-bit).
    K % 64 == 0 (zero padded by the caller)."""
    _gpu(a, b, out)
    Z, M, K = a.shape
    N = b.shape[1]
    assert b.shape[0] == Z and b.shape[2] == K and K % 64 == 0 and tuple(out.shape) == (Z, M, N)
    assert a.stride(2) == 1 and b.stride(2) == 1 and out.stride(2) == 1
    d = GemmDesc()
    d.mode, d.dtype, d.M, d.N, d.K = MODE_LINEAR, _dt(a), M, N, K
    d.A, d.lda, d.strideA_z = a.data_ptr(), a.stride(1), a.stride(0)
    # the kernel uses Kpad as the row stride of the W operand: any stride that is a multiple of 64 and >= K works
    assert b.stride(1) % 64 == 0 and b.stride(1) >= K, "bmm_nt: b row stride must be a multiple of 64 and >= K"
    d.W, d.Wrows, d.Kpad, d.strideW_z = b.data_ptr(), N, b.stride(1), b.stride(0)
    d.act, d.out_scale = ACT_NONE, out_scale
    d.C, d.ldc, d.strideC_z = out.data_ptr(), out.stride(1), out.stride(0)
    d.batch = Z
    _gemm_launch(d, (a, b, out))
    return out


# ------------------------------------------------------------------------------------------------ attention
def attention(q: T, k: T, vt: T, out: T, heads: int, Lk: int, scale: float) -> T:
    """q: [B, Lq, >=H*64] view, k: [B, Lk, >=H*64] view, vt: [B, H*64, Lkpad] (transposed values), out: [B, Lq, H*64]."""
    _gpu(q, k, vt, out)
    B, Lq = q.shape[0], q.shape[1]
    assert q.stride(2) == 1 and k.stride(2) == 1 and vt.stride(2) == 1 and out.stride(2) == 1
    with _Timed("attention", 4.0 * B * heads * Lq * Lk * 64, f"attn B{B} H{heads} Lq{Lq} Lk{Lk}"):
        native.check(native.lib().dbir_attention(
            _dt(q), q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
            vt.data_ptr(), vt.stride(0), vt.stride(1), out.data_ptr(), out.stride(0), out.stride(1),
            B, heads, Lq, Lk, scale, _stream()), "dbir_attention")
    return out


def window_attention(qkv: T, out: T, bias_table: T, C: int, heads: int, ws: int, shift: int, scale: float) -> T:
    """qkv: [B, H, W, >=3C]; out: [B, H, W, >=C]; bias_table f32 [(2ws-1)^2, heads]."""
    _gpu(qkv, out, bias_table)
    B, H, W = qkv.shape[:3]
    native.check(native.lib().dbir_window_attention(
        _dt(qkv), qkv.data_ptr(), _ld(qkv), out.data_ptr(), _ld(out), bias_table.data_ptr(), B, H, W, C, heads, ws,
        shift, scale, _stream()), "dbir_window_attention")
    return out


# ------------------------------------------------------------------------------------------------ fused transformer block
# (csrc/xformer.hip, host side in diffbir_amd/xformer.py).  Counted in the GEMM family for the roofline: the FLOPs are
# those of the linears the kernels replace (bench.py), the LayerNorms / text cross-attention inside are not credited.
from . import xformer as _xf  # noqa: E402

XfBlock = _xf.XfBlock
pack_xf_block = _xf.pack_block
pack_context_frags = _xf.pack_context_frags
xf_supported = _xf.supported


def groupnorm_affine(x: T, gamma: T, beta: T, eps: float, groups: int = 32, stats=None) -> T:
    """GroupNorm(x) as a per (sample, channel) affine map f32 [B, 2, C] for xf_head; `stats`: as in groupnorm()."""
    B, C = x.shape[0], x.shape[-1]
    parts = _usable_partials(stats, B, _rows(x) // B, C)
    if parts is not None:
        return groupnorm_stats_from_partials(parts, B, _rows(x) // B, groups, eps, gamma, beta, affine=True)
    return _xf.groupnorm_affine(x, gamma, beta, eps, groups)



def xf_head(x: T, ab: T, blk: "XfBlock", L: int):
    """GroupNorm-apply -> proj_in -> h; LayerNorm1 -> q | k, v^T (attention.py:344-345, 266, 189-200)."""
    M, C = _rows(x), x.shape[-1]
    with _Timed("gemm", 2.0 * M * C * 4 * C, f"xf_head M{M} C{C}", 2.0 * M * C * 5 + 2.0 * 4 * C * C):
        return _xf.xf_head(x, ab, blk, L)


def xf_tail(attn: T, h: T, x: T, blk: "XfBlock", kf: T, vf: T, Lk: int, scale: float, L: int, out: Optional[T] = None,
            pair_bs: int = 0, stop_after: int = 0) -> T:
    """Everything of the transformer block after the self-attention (attention.py:201-216, 266-273, 19-45, 350-353)."""
    M, C = _rows(attn) * (2 if pair_bs else 1), attn.shape[-1]
    with _Timed("gemm", 2.0 * M * C * 16 * C, f"xf_tail M{M} C{C}", 2.0 * M * C * 4 + 2.0 * 16 * C * C):
        return _xf.xf_tail(attn, h, x, blk, kf, vf, Lk, scale, L, out=out, pair_bs=pair_bs, stop_after=stop_after)


# ------------------------------------------------------------------------------------------------ norms
def _usable_partials(stats, B: int, HW: int, C: int):
    """stats: GnPartials, or a (left, right) pair for a two-producer concat buffer -> (p1, p2 | None) when the column sums
    cover the tensor exactly in whole per-sample tiles, else None (the caller computes the statistics itself)."""
    if stats is None:
        return None
    parts = list(stats) if isinstance(stats, (tuple, list)) else [stats]
    if not parts or len(parts) > 2 or any(p is None for p in parts):
        return None
    if any(p.rows != parts[0].rows or p.M != B * HW for p in parts) or HW % parts[0].rows or sum(p.N for p in parts) != C:
        return None
    return parts[0], (parts[1] if len(parts) == 2 else None)


def groupnorm_stats_from_partials(parts, B: int, HW: int, groups: int, eps: float, gamma: Optional[T] = None,
                                  beta: Optional[T] = None, affine: bool = False) -> T:
    """(p1, p2 | None) -> mean_var f32 [B, 2 * groups], or with affine=True the scale | shift map f32 [B, 2, C]."""
    p1, p2 = parts
    C = p1.N + (p2.N if p2 is not None else 0)
    dev = p1.buf.device
    mv = None if affine else torch.empty((B, 2 * groups), dtype=torch.float32, device=dev)
    ab = torch.empty((B, 2, C), dtype=torch.float32, device=dev) if affine else None
    native.check(native.lib().dbir_groupnorm_from_partials(
        p1.buf.data_ptr(), p1.N, None if p2 is None else p2.buf.data_ptr(), 0 if p2 is None else p2.N, p1.rows, B, HW,
        groups, eps, None if gamma is None else gamma.data_ptr(), None if beta is None else beta.data_ptr(),
        None if mv is None else mv.data_ptr(), None if ab is None else ab.data_ptr(), _stream()),
        "dbir_groupnorm_from_partials")
    return ab if affine else mv


# the statistics merge of the epilogue partials inside the normalising kernel (dbir_groupnorm_apply_partials) instead of a
# launch of its own in front of it.  MEASURED SLOWER and therefore OFF (DBIR_GN_FUSED_PARTIALS=1 switches it on): every block
# of the normalising launch repeats the merge of its sample's ~5000 scattered (sum, M2) cells in two dependent passes before
# its first row — 6.69 img/s with the separate 5 us merge launch vs 6.47 - 6.53 fused, same box, interleaved
# (profiles/r5_call1_ab.txt).  Kept, with its parity test, as the measured counter-example to "fold every small launch".
GN_FUSED_PARTIALS = os.environ.get("DBIR_GN_FUSED_PARTIALS", "0") == "1"


def groupnorm(x: T, gamma: T, beta: T, eps: float, silu: bool, out: Optional[T] = None, groups: int = 32,
              stats=None) -> T:
    """x: [B, H, W, C] (or [B, HW, C]) 16-bit; gamma/beta f32 [C].
    stats: the producing GEMM's epilogue column sums (GnPartials, or a (left, right) pair for a concat buffer): the
    statistics pass over x is skipped (GroupNorm statistics from the producer)."""
    _gpu(x, gamma, beta, out)
    B, C = x.shape[0], x.shape[-1]
    HW = _rows(x) // B
    parts = _usable_partials(stats, B, HW, C)
    if parts is not None and GN_FUSED_PARTIALS:
        p1, p2 = parts
        if out is None:
            out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        native.check(native.lib().dbir_groupnorm_apply_partials(
            _dt(x), x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), gamma.data_ptr(), beta.data_ptr(), p1.buf.data_ptr(), p1.N,
            None if p2 is None else p2.buf.data_ptr(), 0 if p2 is None else p2.N, p1.rows, B, HW, groups, eps, int(silu),
            _stream()), "dbir_groupnorm_apply_partials")
        return out
    if parts is not None:
        mv = groupnorm_stats_from_partials(parts, B, HW, groups, eps)
        return groupnorm_apply(x, gamma, beta, mv, eps, silu, out=out, groups=groups)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    nchunk = native.lib().dbir_groupnorm_nchunk(HW, C)
    ws = torch.empty(B * (2 * C * nchunk + 2 * C), dtype=torch.float32, device=x.device)
    native.check(native.lib().dbir_groupnorm(
        _dt(x), x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), gamma.data_ptr(), beta.data_ptr(), B, HW, C, groups,
        eps, int(silu), ws.data_ptr(), _stream()), "dbir_groupnorm")
    return out


def groupnorm_stats(x: T, groups: int = 32) -> T:
    """Per-sample GroupNorm statistics of x [B, H, W, C]: f32 [B, 2*groups] = mean(groups) | biased variance(groups)."""
    _gpu(x)
    B, C = x.shape[0], x.shape[-1]
    HW = _rows(x) // B
    nchunk = native.lib().dbir_groupnorm_nchunk(HW, C)
    ws = torch.empty(B * (2 * C * nchunk + 2 * C), dtype=torch.float32, device=x.device)
    mv = torch.empty((B, 2 * groups), dtype=torch.float32, device=x.device)
    native.check(native.lib().dbir_groupnorm_stats(_dt(x), x.data_ptr(), _ld(x), B, HW, C, groups, ws.data_ptr(),
                                                   mv.data_ptr(), _stream()), "dbir_groupnorm_stats")
    return mv


def groupnorm_apply(x: T, gamma: T, beta: T, mean_var: T, eps: float, silu: bool, out: Optional[T] = None,
                    groups: int = 32) -> T:
    """GroupNorm(+SiLU) of x with caller-supplied statistics `mean_var` f32 [B, 2*groups] (tiled VAE)."""
    _gpu(x, gamma, beta, mean_var, out)
    B, C = x.shape[0], x.shape[-1]
    HW = _rows(x) // B
    assert mean_var.dtype == torch.float32 and mean_var.is_contiguous() and tuple(mean_var.shape) == (B, 2 * groups)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    native.check(native.lib().dbir_groupnorm_apply(_dt(x), x.data_ptr(), _ld(x), out.data_ptr(), _ld(out),
                                                   gamma.data_ptr(), beta.data_ptr(), mean_var.data_ptr(), B, HW, C,
                                                   groups, eps, int(silu), _stream()), "dbir_groupnorm_apply")
    return out


def layernorm(x: T, gamma: T, beta: T, C: Optional[int] = None, eps: float = 1e-5, out: Optional[T] = None) -> T:
    """Row LayerNorm over the first C columns of x [..., Cpad]; pad columns of the output are zero."""
    _gpu(x, gamma, beta, out)
    Cpad = x.shape[-1]
    C = Cpad if C is None else C
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    native.check(native.lib().dbir_layernorm(
        _dt(x), x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), gamma.data_ptr(), beta.data_ptr(), _rows(x), C, Cpad,
        eps, _stream()), "dbir_layernorm")
    return out


def softmax_rows_(x: T, L: int) -> T:
    """In-place softmax over the first L columns of every row of x [..., ld]; the rest is zeroed."""
    _gpu(x)
    assert x.is_contiguous()
    native.check(native.lib().dbir_softmax_rows(_dt(x), x.data_ptr(), x.shape[-1], _rows(x), L, _stream()),
                 "dbir_softmax_rows")
    return x


# ------------------------------------------------------------------------------------------------ CLIP text tower
def clip_embed(tokens: T, tok_emb: T, pos: T) -> T:
    """tokens int64 [B, L]; tok_emb f32 [vocab, W]; pos f32 [L, W] -> f32 [B, L, W] (token + positional embedding)."""
    _gpu(tokens, tok_emb, pos)
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and tok_emb.is_contiguous() and pos.is_contiguous()
    assert tok_emb.dtype == torch.float32 and pos.dtype == torch.float32
    B, L = tokens.shape
    W = tok_emb.shape[1]
    assert pos.shape[0] >= L and pos.shape[1] == W
    x = torch.empty((B, L, W), dtype=torch.float32, device=tokens.device)
    native.check(native.lib().dbir_clip_embed(tokens.data_ptr(), tok_emb.data_ptr(), pos.data_ptr(), x.data_ptr(), B, L,
                                              W, tok_emb.shape[0], _stream()), "dbir_clip_embed")
    return x


def add_layernorm_f32(x: T, y: Optional[T], gamma: T, beta: T, out_dtype: torch.dtype, eps: float = 1e-5) -> T:
    """x f32 [..., C] += y (f32 or None) IN PLACE; returns LayerNorm(x) * gamma + beta in `out_dtype` (16-bit: the next
    GEMM's operand; f32: the tower's output)."""
    _gpu(x, y, gamma, beta)
    assert x.dtype == torch.float32 and x.is_contiguous() and (y is None or (y.dtype == torch.float32 and y.is_contiguous()
                                                                              and y.shape == x.shape))
    C = x.shape[-1]
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    f32 = out_dtype == torch.float32
    native.check(native.lib().dbir_add_layernorm_f32(
        native.F16 if f32 else (native.BF16 if out_dtype == torch.bfloat16 else native.F16), x.data_ptr(), None if y is None else y.data_ptr(), gamma.data_ptr(),
        beta.data_ptr(), out.data_ptr(), C, int(f32), _rows(x), C, eps, _stream()), "dbir_add_layernorm_f32")
    return out


def causal_attention(qkv: T, heads: int, scale: float) -> T:
    """qkv 16-bit [B, L, 3*heads*64] (q | k | v) -> [B, L, heads*64]; causal mask, L <= 128."""
    _gpu(qkv)
    assert qkv.stride(2) == 1 and qkv.stride(0) == qkv.shape[1] * qkv.stride(1)
    B, L = qkv.shape[:2]
    out = torch.empty((B, L, heads * 64), dtype=qkv.dtype, device=qkv.device)
    native.check(native.lib().dbir_causal_attention(_dt(qkv), qkv.data_ptr(), qkv.stride(1), out.data_ptr(), heads * 64, B,
                                                    heads, L, scale, _stream()), "dbir_causal_attention")
    return out


# ------------------------------------------------------------------------------------------------ elementwise
def add_scaled(a: T, b: T, s: float, out: Optional[T] = None) -> T:
    _gpu(a, b, out)
    if out is None:
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    native.check(native.lib().dbir_add_scaled(_dt(a), a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), s, out.data_ptr(),
                                              _ld(out), _rows(a), a.shape[-1], _stream()), "dbir_add_scaled")
    return out


def copy_rows(src: T, dst: T) -> T:
    """dst[..., :C] = src[..., :C] for 16-bit channels-last tensors of equal shape (row strides may differ)."""
    _gpu(src, dst)
    assert src.shape == dst.shape and src.dtype == dst.dtype and src.element_size() == 2
    native.check(native.lib().dbir_copy_rows(src.data_ptr(), _ld(src), dst.data_ptr(), _ld(dst), _rows(src), src.shape[-1],
                                             _stream()), "dbir_copy_rows")
    return dst


def copy_rows2d(src: T, dst: T) -> T:
    """dst[m, :] = src[m, :] for 2-D tensors of ANY element type whose rows are contiguous (row strides may differ); a row must
    be a multiple of 16 bytes.  (dbir_copy_rows moves 16-byte chunks: the row is described in 2-byte units.)"""
    _gpu(src, dst)
    assert src.dim() == 2 and src.shape == dst.shape and src.dtype == dst.dtype and src.stride(1) == 1 and dst.stride(1) == 1
    es = src.element_size()
    rb, sb, db = src.shape[1] * es, src.stride(0) * es, dst.stride(0) * es
    assert rb % 16 == 0 and sb % 16 == 0 and db % 16 == 0, "copy_rows2d: rows and row strides must be multiples of 16 bytes"
    native.check(native.lib().dbir_copy_rows(src.data_ptr(), sb // 2, dst.data_ptr(), db // 2, src.shape[0], rb // 2, _stream()),
                 "dbir_copy_rows")
    return dst


def space_to_depth2(x: T) -> T:
    """[B, 2h, 2w, C] 16-bit NHWC (row stride may exceed C) -> [B, h, w, 4C], channel = (ky*2 + kx)*C + c."""
    _gpu(x)
    B, H, W, C = x.shape
    assert H % 2 == 0 and W % 2 == 0
    out = torch.empty((B, H // 2, W // 2, 4 * C), dtype=x.dtype, device=x.device)
    native.check(native.lib().dbir_block2x2(x.data_ptr(), _ld(x), out.data_ptr(), 4 * C, B, H // 2, W // 2, C, 1,
                                            _stream()), "dbir_block2x2")
    return out


def depth_to_space2(x: T, out: Optional[T] = None) -> T:
    """[B, h, w, 4C] -> [B, 2h, 2w, C] (inverse of space_to_depth2); `out` may be a column slice of a wider buffer."""
    _gpu(x, out)
    B, h, w, C4 = x.shape
    C = C4 // 4
    if out is None:
        out = torch.empty((B, 2 * h, 2 * w, C), dtype=x.dtype, device=x.device)
    native.check(native.lib().dbir_block2x2(x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), B, h, w, C, 0, _stream()),
                 "dbir_block2x2")
    return out


def nchw_to_nhwc(src0: T, src1: Optional[T], cpad: int, dtype, scale: float = 1.0, shift: float = 0.0) -> T:
    _gpu(src0, src1)
    assert src0.dtype == torch.float32 and src0.is_contiguous() and (src1 is None or src1.is_contiguous())
    B, C0, H, W = src0.shape
    C1 = 0 if src1 is None else src1.shape[1]
    out = torch.empty((B, H, W, cpad), dtype=dtype, device=src0.device)
    native.check(native.lib().dbir_nchw_to_nhwc(_dt(out), src0.data_ptr(), C0, None if src1 is None else src1.data_ptr(),
                                                C1, out.data_ptr(), cpad, B, H, W, scale, shift, _stream()),
                 "dbir_nchw_to_nhwc")
    return out


def nhwc_to_nchw(src: T, C: int, scale: float = 1.0, shift: Optional[T] = None) -> T:
    _gpu(src, shift)
    B, H, W = src.shape[:3]
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=src.device)
    f32 = src.dtype == torch.float32
    native.check(native.lib().dbir_nhwc_to_nchw(native.F16 if f32 else _dt(src), src.data_ptr(), int(f32), _ld(src),
                                                out.data_ptr(), C, B, H, W, scale,
                                                None if shift is None else shift.data_ptr(), _stream()),
                 "dbir_nhwc_to_nchw")
    return out


def pixel_unshuffle(src: T, r: int, cpad: int, mean: T, rng: float, dtype) -> T:
    _gpu(src, mean)
    assert src.dtype == torch.float32 and src.is_contiguous()
    B, C, H, W = src.shape
    out = torch.empty((B, H // r, W // r, cpad), dtype=dtype, device=src.device)
    native.check(native.lib().dbir_pixel_unshuffle(_dt(out), src.data_ptr(), out.data_ptr(), B, C, H, W, r, cpad,
                                                   mean.data_ptr(), rng, _stream()), "dbir_pixel_unshuffle")
    return out


def timestep_embedding(t: T, dim: int, dtype, max_period: float = 10000.0) -> T:
    _gpu(t)
    t = t.to(torch.float32).contiguous()
    out = torch.empty((t.shape[0], dim), dtype=dtype, device=t.device)
    native.check(native.lib().dbir_timestep_embedding(_dt(out), t.data_ptr(), out.data_ptr(), t.shape[0], dim,
                                                      max_period, _stream()), "dbir_timestep_embedding")
    return out


def lincomb4(x: T, ca: T, y: Optional[T] = None, cb: Optional[T] = None, z: Optional[T] = None,
             cc: Optional[T] = None, w: Optional[T] = None, cd: Optional[T] = None) -> T:
    """out = ca[b]*x + cb[b]*y + cc[b]*z + cd[b]*w on f32 tensors [B, ...]; coefficient tensors f32 [B]."""
    _gpu(x, y, z, w, ca, cb, cc, cd)
    for t in (x, y, z, w):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    B = x.shape[0]
    out = torch.empty_like(x)
    p = lambda t: None if t is None else t.data_ptr()
    native.check(native.lib().dbir_lincomb4(p(x), p(y), p(z), p(w), p(ca), p(cb), p(cc), p(cd), out.data_ptr(), B,
                                            x.numel() // B, _stream()), "dbir_lincomb4")
    return out


def spaced_step(x: T, oc: T, ou: Optional[T], noise: T, s: float, k_x: T, k_o: T, c1: T, c2: T, sd: T) -> T:
    _gpu(x, oc, ou, noise, k_x, k_o, c1, c2, sd)
    for t in (x, oc, ou, noise):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    B = x.shape[0]
    out = torch.empty_like(x)
    native.check(native.lib().dbir_spaced_step(x.data_ptr(), oc.data_ptr(), None if ou is None else ou.data_ptr(),
                                               noise.data_ptr(), s, k_x.data_ptr(), k_o.data_ptr(), c1.data_ptr(),
                                               c2.data_ptr(), sd.data_ptr(), out.data_ptr(), B, x.numel() // B,
                                               _stream()), "dbir_spaced_step")
    return out


def tile_gather(x: T, coords: T, ts: int) -> T:
    """x f32 [B,C,H,W], coords int32 [T,2] -> tiles f32 [T*B, C, ts, ts] (tile-major)."""
    _gpu(x, coords)
    assert x.dtype == torch.float32 and x.is_contiguous() and coords.dtype == torch.int32 and coords.is_contiguous()
    B, C, H, W = x.shape
    Tn = coords.shape[0]
    out = torch.empty((Tn * B, C, ts, ts), dtype=torch.float32, device=x.device)
    native.check(native.lib().dbir_tile_gather(x.data_ptr(), out.data_ptr(), coords.data_ptr(), Tn, B, C, H, W, ts,
                                               _stream()), "dbir_tile_gather")
    return out


def tile_accumulate(tiles: T, weights: T, coords: T, B: int, H: int, W: int) -> T:
    _gpu(tiles, weights, coords)
    assert tiles.dtype == torch.float32 and tiles.is_contiguous() and weights.dtype == torch.float32
    Tn = coords.shape[0]
    C, ts = tiles.shape[1], tiles.shape[2]
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=tiles.device)
    native.check(native.lib().dbir_tile_accumulate(tiles.data_ptr(), weights.data_ptr(), coords.data_ptr(),
                                                   out.data_ptr(), Tn, B, C, H, W, ts, _stream()),
                 "dbir_tile_accumulate")
    return out


def tile_accumulate_partial(tiles: Optional[T], weights: T, coords: T, B: int, C: int, H: int, W: int) -> T:
    """Un-normalised weighted sum over the tiles given (a rank's shard); tiles=None -> the normaliser (B=C=1)."""
    _gpu(tiles, weights, coords)
    assert weights.dtype == torch.float32 and coords.dtype == torch.int32 and coords.is_contiguous()
    Tn, ts = coords.shape[0], weights.shape[-1]
    if tiles is not None:
        assert tiles.dtype == torch.float32 and tiles.is_contiguous() and tuple(tiles.shape) == (Tn * B, C, ts, ts)
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=weights.device)
    native.check(native.lib().dbir_tile_accumulate_partial(None if tiles is None else tiles.data_ptr(),
                                                           weights.data_ptr(), coords.data_ptr(), out.data_ptr(), Tn,
                                                           B, C, H, W, ts, _stream()), "dbir_tile_accumulate_partial")
    return out


def tile_normalize(num: T, den: T) -> T:
    """out[b,c] = num[b,c] / den  (den: f32 [H,W] or [1,1,H,W])."""
    _gpu(num, den)
    assert num.dtype == torch.float32 and num.is_contiguous() and den.is_contiguous() and den.dtype == torch.float32
    H, W = num.shape[-2:]
    assert den.numel() == H * W
    out = torch.empty_like(num)
    native.check(native.lib().dbir_tile_normalize(num.data_ptr(), den.data_ptr(), out.data_ptr(),
                                                  num.numel() // (H * W), H * W, _stream()), "dbir_tile_normalize")
    return out


def u8_to_f32_nchw(src: T) -> T:
    _gpu(src)
    assert src.dtype == torch.uint8 and src.is_contiguous() and src.shape[-1] == 3
    B, H, W = src.shape[:3]
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=src.device)
    native.check(native.lib().dbir_u8_to_f32_nchw(src.data_ptr(), out.data_ptr(), B, H, W, _stream()),
                 "dbir_u8_to_f32_nchw")
    return out


def wavelet_blur(src: T, radius: int) -> T:
    _gpu(src)
    assert src.dtype == torch.float32 and src.is_contiguous()
    H, W = src.shape[-2:]
    out = torch.empty_like(src)
    native.check(native.lib().dbir_wavelet_blur(src.data_ptr(), out.data_ptr(), src.numel() // (H * W), H, W, radius,
                                                _stream()), "dbir_wavelet_blur")
    return out


def colorfix(content: T, content_low: T, style_low: T) -> T:
    _gpu(content, content_low, style_low)
    out = torch.empty_like(content)
    native.check(native.lib().dbir_colorfix(content.data_ptr(), content_low.data_ptr(), style_low.data_ptr(),
                                            out.data_ptr(), content.numel(), _stream()), "dbir_colorfix")
    return out


def f32_nchw_to_u8_nhwc(src: T) -> T:
    _gpu(src)
    assert src.dtype == torch.float32 and src.is_contiguous() and src.shape[1] == 3
    B, _, H, W = src.shape
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=src.device)
    native.check(native.lib().dbir_f32_nchw_to_u8_nhwc(src.data_ptr(), out.data_ptr(), B, H, W, _stream()),
                 "dbir_f32_nchw_to_u8_nhwc")
    return out
