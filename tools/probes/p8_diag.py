#!/usr/bin/env python
"""(needs a `DBIR_DIAG=1 sh diffbir_amd/csrc/build.sh` library)  Anatomy of the fine-phase kernel (tile 80) on one convolution: per-section s_memtime accumulators (DBIR_P8_VAR=32/34)
and compile-time ablations (2 no setprio, 4 no MFMA, 8 no staging, 16 no fragment reads, 64 lgkmcnt before the barrier).
Each variant needs its own process (the variant is read once): python tools/probes/p8_diag.py <var> [B H W Cin Cout]"""
import os
import sys

var = sys.argv[1] if len(sys.argv) > 1 else "0"
os.environ["DBIR_P8_VAR"] = var
os.environ["DBIR_AUTOTUNE"] = "0"
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffbir_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
DT = torch.float16
b, h, w, ci, co = [int(v) for v in sys.argv[2:7]] if len(sys.argv) >= 7 else (16, 64, 64, 320, 320)
ws = torch.zeros(1 << 20, dtype=torch.int64, device=DEV)
_orig = ops.apply_tile_code


def _patched(d, code, device):
    _orig(d, code, device)
    if d.splitk <= 1:
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 8


ops.apply_tile_code = _patched
x = torch.randn(b, h, w, ci, device=DEV).to(DT)
pw = ops.pack_conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DT, DEV)
r = None if os.environ.get("P8_NORES") else torch.randn(b, h, w, co, device=DEV).to(DT)
emb = None if os.environ.get("P8_NOEMB") else torch.randn(b, co, device=DEV).to(DT)
out = torch.empty(b, h, w, co, dtype=DT, device=DEV)
fl = 2.0 * b * h * w * co * 9 * ci
best = 1e9
for rep in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ops.conv3x3(x, pw, residual=r, rowvec=emb, out=out, tile=80)
    e1.record()
    e1.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / 4)
line = f"var {var:>3s}{' nores' if r is None else ''}{' noemb' if emb is None else ''}: conv B{b} {h}x{w} {ci}->{co}: {best:7.1f} us  {fl / best * 1e-6:5.0f} TF/s"
if int(var) & 32:
    torch.cuda.synchronize()
    nb = ((b * h * w + 255) // 256) * ((co + 319) // 320)
    t = ws[: nb * 64].reshape(nb, 8, 8).double().cpu()
    for g in (0, 1):
        sel = t[:, 4 * g:4 * g + 4].reshape(-1, 8)
        m = sel.mean(0)
        nph = 2 * m[5]
        line += (f"\n   group {g}: per phase (cycles @100MHz ticks x ?): R {m[0] / nph:7.1f}  barrier1 {m[1] / nph:7.1f}  "
                 f"M {m[2] / nph:7.1f}  barrier2 {m[3] / nph:7.1f}   loop total {m[4]:9.0f} ticks, {int(m[5])} stages; "
                 f"s_memrealtime {m[7]:7.0f} x 10 ns -> s_memtime runs at {m[4] / m[7] * 100:6.0f} MHz")
    t2 = ws[64 * 2048: 64 * 2048 + nb * 64].reshape(nb, 8, 8).double().cpu()
    line += (f"\n   per wave (x 10 ns): prologue {t2[..., 0].mean():6.0f}  loop {t[..., 7].mean():6.0f}  "
             f"whole kernel {t2[..., 1].mean():6.0f} (max {t2[..., 1].max():6.0f}); kernel start -> table built {t2[..., 2].mean():6.0f}, "
             f"pass 0 {t2[..., 3].mean():6.0f}, pass 1 + drain {t2[..., 4].mean():6.0f}")
print(line, flush=True)
