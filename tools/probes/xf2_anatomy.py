"""Section anatomy of the second-generation xf_tail (GPU, DBIR_DIAG build): s_memtime per section (stop_after = 99), both wave groups.
DBIR_DIAG=1 sh diffbir_amd/csrc/build.sh && python tools/probes/xf2_anatomy.py [C ...]"""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_kernels_gpu import _xf_weights
from diffbir_amd import ops

DEV = torch.device("cuda:0")
dtype = torch.float16
names = ["panel load+init", "GEMM out1", "round+LN2+bias", "GEMM q2", "store q", "cross-attn+init", "GEMM out2", "LN3+init",
         "feed-forward", "h3->X, res load", "GEMM proj_out", "row store"]
for C in [int(a) for a in sys.argv[1:]] or [320, 640]:
    L, Lk, B = (4096, 77, 16) if C == 320 else (1024, 77, 16)
    blk = ops.pack_xf_block(_xf_weights(C=C), dtype, DEV)
    M = B * L
    attn, h = torch.randn(M, C, device=DEV).to(dtype), torch.randn(M, C, device=DEV).to(dtype)
    side = 64 if C == 320 else 32
    x = torch.randn(B, side, side, C, device=DEV).to(dtype)
    k, vt = torch.randn(B, Lk, C, device=DEV).to(dtype), torch.randn(B, C, 80, device=DEV).to(dtype)
    kf, vf = ops.pack_context_frags(k, vt, Lk, C // 64)
    out = torch.zeros_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        hh = h.clone()   # the timing build writes its section table over the first 64 KB of `h`
        e0.record()
        ops.xf_tail(attn, hh, x, blk, kf, vf, Lk, 0.125, L, out=out, stop_after=99)
        e1.record()
    torch.cuda.synchronize()
    print(f"C {C} M {M}: kernel wall {e0.elapsed_time(e1) * 1e3:.1f} us")
    ta = hh.view(torch.int64).flatten()[: 256 * 2 * 16].reshape(256, 2, 16).double()
    for g in range(2):
        mean = ta[:, g].mean(0)
        tot = mean[:12].sum().item()
        print(f" wave group {g}: total {tot:.0f} ticks per workgroup")
        for i, n in enumerate(names):
            print(f"  {n:22s} {mean[i].item():10.0f}  {mean[i].item() / tot * 100:5.1f}%   (min {ta[:, g, i].min().item():.0f} max {ta[:, g, i].max().item():.0f})")
