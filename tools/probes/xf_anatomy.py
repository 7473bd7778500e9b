"""Section anatomy of xf_tail (GPU): s_memtime per section from the debug instantiation (stop_after = 99)."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_kernels_gpu import _xf_weights
from diffbir_amd import native, ops

DEV = torch.device("cuda:0")
dtype = torch.float16
C, L, Lk, B = 320, 4096, 77, 8
blk = ops.pack_xf_block(_xf_weights(), dtype, DEV)
M = B * L
attn, h = torch.randn(M, C, device=DEV).to(dtype), torch.randn(M, C, device=DEV).to(dtype)
x = torch.randn(B, 64, 64, C, device=DEV).to(dtype)
k, vt = torch.randn(B, Lk, C, device=DEV).to(dtype), torch.randn(B, C, 80, device=DEV).to(dtype)
kf, vf = ops.pack_context_frags(k, vt, Lk, 5)
names = ["panel load+init", "GEMM out1", "round+LN2", "GEMM q2", "store q", "cross-attn", "init out2", "GEMM out2",
         "LN3+init", "FF1 tiles (x20)", "GELU+g store (x20)", "FF2 tiles (x20)", "h3->X, res load", "GEMM proj_out", "row store"]
for v in [int(a) for a in sys.argv[1:]] or [1]:
    native.check(native.lib().dbir_set_option(2, v), "set_option")
    out = torch.zeros_like(x)
    hh = h.clone()   # the timing build writes its section table over the first 32 KB of `h`
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        e0.record()
        ops.xf_tail(attn, hh, x, blk, kf, vf, Lk, 0.125, L, out=out, stop_after=99)
        e1.record()
    torch.cuda.synchronize()
    print(f"kernel wall {e0.elapsed_time(e1) * 1e3:.1f} us")
    ta = hh.view(torch.int64).flatten()[: 256 * 16].reshape(256, 16).double()
    mean = ta.mean(0)
    tot = mean.sum().item()
    print(f"variant {v}: total {tot:.0f} ticks per workgroup (1 panel each)")
    for i, n in enumerate(names):
        print(f"  {n:22s} {mean[i].item():10.0f}  {mean[i].item() / tot * 100:5.1f}%   (min {ta[:, i].min().item():.0f} max {ta[:, i].max().item():.0f})")


def wall(code, n=6):
    best = 1e9
    hh = h.clone()
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.xf_tail(attn, hh, x, blk, kf, vf, Lk, 0.125, L, out=out, stop_after=code)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3


print(f"wall-clock ablations (B{B}, one panel per workgroup): full {wall(0):.1f} us | no staging {wall(103):.1f} | "
      f"no MFMAs {wall(104):.1f} | no fragment reads {wall(105):.1f}")
