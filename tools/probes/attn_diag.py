#!/usr/bin/env python
"""DIAG build only (DBIR_DIAG=1 sh diffbir_amd/csrc/build.sh): per-wave s_memtime anatomy of the attention tile loop."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffbir_amd import ops
DEV = torch.device("cuda:0")
B, H, L = 16, 5, 4096
C = H * 64
q = torch.randn(B, L, C, device=DEV).half(); k = torch.randn(B, L, C, device=DEV).half()
vt = torch.randn(B, C, L, device=DEV).half(); o = torch.zeros(B, L, C, device=DEV).half()
ops.attention(q, k, vt, o, H, L, 0.125)
torch.cuda.synchronize()
rows = o.view(B, L // 32, 32, C)[:, :, 0, :].contiguous()           # first row of every wave's 32-row slab
v = rows.view(B, L // 32, H, 64)[..., :28].contiguous().view(torch.int64).reshape(-1, 7).double()
v = v[v[:, 6] > 0]
nt = v[:, 6].mean().item()
names = ["fetch issue", "QK^T MFMA (+LDS reads)", "softmax VALU", "PV MFMA (+LDS reads)", "commit (vmcnt + LDS writes)", "barrier"]
print(f"waves {v.shape[0]}, tiles {nt:.0f}; cycles per 64-key tile:")
for i, n in enumerate(names):
    print(f"  {n:30s} {v[:, i].mean().item() / nt:8.0f}")
print(f"  {'total':30s} {v[:, :6].sum(1).mean().item() / nt:8.0f}   (MFMA pipe work per wave: 512)")
