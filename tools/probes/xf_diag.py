"""Diagnostic (GPU): error map of xf_head / xf_tail against the f32 statement."""
import sys
import torch
sys.path.insert(0, ".")
from tests import emu_ops as emu
from tests.test_kernels_gpu import _xf_weights
from diffbir_amd import ops

DEV = torch.device("cuda:0")
torch.manual_seed(0)
for dtype in (torch.float16, torch.bfloat16):
    for B, L in ((2, 128), (3, 4096)):
        C = 320
        blk = ops.pack_xf_block(_xf_weights(seed=1), dtype, DEV)
        x = (torch.randn(B, L // 64, 64, C, device=DEV) * 1.5 + 0.3).to(dtype)
        gam, bet = 1 + 0.1 * torch.randn(C, device=DEV), 0.1 * torch.randn(C, device=DEV)
        ab = ops.groupnorm_affine(x, gam, bet, 1e-6)
        outs = [ops.xf_head(x, ab, blk, L) for _ in range(3)]
        ref = emu.xf_head(x, ab, blk, L)
        for name, i in (("h", 0), ("qk", 1), ("vt", 2)):
            g, r = outs[0][i].float(), ref[i].float()
            err = (g - r).abs()
            bad = err > 0.05 * r.abs().max()
            same = all(torch.equal(outs[0][i], o[i]) for o in outs[1:])
            print(f"{dtype} B{B} L{L} {name}: rel_l2={(g - r).norm() / r.norm():.3e} bad={int(bad.sum())}/{bad.numel()} reproducible={same}")
            if bad.any() and name == "h":
                rows = bad.any(dim=1).nonzero().flatten()
                cols = bad.any(dim=0).nonzero().flatten()
                print("   bad rows:", rows[:40].tolist(), "... n =", len(rows))
                print("   bad cols:", cols[:64].tolist(), "... n =", len(cols))
                rr = rows[0].item()
                print("   row", rr, "got", g[rr, :12].tolist(), "\n          ref", r[rr, :12].tolist())
