#!/usr/bin/env python
"""Per-kernel microbenchmark on one MI355X: the hot-path shapes of one batched network evaluation (2B = 16 samples,
64x64 latent) and of the VAE, timed per launch with HIP events on the launch stream; prints achieved TFLOP/s (MFMA
kernels) or GB/s (HBM-bound kernels) per tile variant.  Usage: python tools/bench_kernels.py [--quick] [--json out]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffbir_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
DT = torch.float16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def rnd(*shape, s=1.0):
    return (torch.randn(*shape, device=DEV) * s).to(DT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--json", default=None)
    ap.add_argument("--tiles", default="1,5,6,7")
    ap.add_argument("--only", default="")
    ap.add_argument("--attn-variants", default="2,3", help="comma list of attention kernel variants to time")
    args = ap.parse_args()
    tiles = [int(t) for t in args.tiles.split(",")]
    rows = []

    def report(name, variant, sec, flops=None, bytes_=None):
        r = dict(name=name, variant=variant, us=sec * 1e6)
        if flops:
            r["tflops"] = flops / sec / 1e12
        if bytes_:
            r["gbps"] = bytes_ / sec / 1e9
        rows.append(r)
        print(f"{name:44s} {variant:>8s} {sec*1e6:10.1f} us " + (f"{r.get('tflops', 0):8.1f} TF/s" if flops else "") +
              (f"{r.get('gbps', 0):8.0f} GB/s" if bytes_ else ""), flush=True)

    want = lambda k: (not args.only) or any(o in k for o in args.only.split(","))
    B = 16
    if want("conv"):
        convs = [(B, 64, 64, 320, 320, 1, False), (B, 32, 32, 640, 640, 1, False), (B, 16, 16, 1280, 1280, 1, False),
                 (B, 8, 8, 1280, 1280, 1, False), (B, 16, 16, 2560, 1280, 1, False), (B, 64, 64, 960, 320, 1, False),
                 (B, 32, 32, 1920, 640, 1, False), (B, 64, 64, 320, 320, 2, False), (B, 32, 32, 640, 640, 1, True),
                 (8, 512, 512, 128, 128, 1, False), (8, 256, 256, 256, 256, 1, False), (8, 128, 128, 512, 512, 1, False),
                 (8, 64, 64, 512, 512, 1, False)]
        if args.quick:
            convs = convs[:4]
        for (b, h, w, ci, co, st, up) in convs:
            x = rnd(b, h, w, ci)
            pw = ops.pack_conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DT, DEV)
            ho = (2 * h if up else h) // st
            wo = (2 * w if up else w) // st
            out = torch.empty(b, ho, wo, co, dtype=DT, device=DEV)
            fl = 2.0 * b * ho * wo * co * 9 * ci
            for t in tiles:
                try:
                    sec = timeit(lambda: ops.conv3x3(x, pw, stride=st, upsample=up, out=out, tile=t))
                    report(f"conv3x3 {b}x{h}x{w} {ci}->{co} s{st} u{int(up)}", f"t{t}", sec, flops=fl)
                except Exception as e:  # noqa: BLE001
                    print("skip", t, str(e)[:80])
            del x, out, pw
    if want("linear"):
        lins = [(65536, 320, 320, False), (65536, 2560, 320, True), (65536, 320, 1280, False), (16384, 640, 640, False),
                (16384, 5120, 640, True), (16384, 640, 2560, False), (4096, 1280, 1280, False), (4096, 10240, 1280, True),
                (4096, 1280, 5120, False), (1024, 1280, 1280, False), (65536, 960, 320, False), (1232, 1280, 1024, False)]
        if args.quick:
            lins = lins[:3]
        for (M, N, K, geglu) in lins:
            x = rnd(M, K)
            if geglu:
                pw = ops.pack_geglu(torch.randn(2 * N, K) * K ** -0.5, torch.randn(2 * N), DT, DEV)
            else:
                pw = ops.pack_linear(torch.randn(N, K) * K ** -0.5, torch.randn(N), DT, DEV)
            out = torch.empty(M, N, dtype=DT, device=DEV)
            fl = 2.0 * M * (2 * N if geglu else N) * K
            for t in tiles:
                if geglu and t in (3, 4):
                    continue
                try:
                    sec = timeit(lambda: ops.linear(x, pw, out=out, tile=t))
                    report(f"linear {M}x{N}x{K}{' geglu' if geglu else ''}", f"t{t}", sec, flops=fl)
                except Exception as e:  # noqa: BLE001
                    print("skip", t, str(e)[:80])
            del x, out, pw
    if want("attn"):
        for (b, hds, lq, lk) in [(B, 5, 4096, 4096), (B, 10, 1024, 1024), (B, 20, 256, 256), (B, 5, 4096, 77),
                                 (B, 10, 1024, 77), (B, 20, 256, 77), (B, 20, 64, 77)]:
            C = hds * 64
            q, k = rnd(b, lq, C), rnd(b, lk, C)
            lkp = (lk + 7) // 8 * 8
            vt = rnd(b, C, lkp)
            o = torch.empty(b, lq, C, dtype=DT, device=DEV)
            from diffbir_amd import native
            for v in [int(x) for x in args.attn_variants.split(",")]:
                native.check(native.lib().dbir_set_option(1, v), "dbir_set_option")
                sec = timeit(lambda: ops.attention(q, k, vt, o, hds, lk, 0.125))
                report(f"attention B{b} H{hds} Lq{lq} Lk{lk}", f"v{v}", sec, flops=4.0 * b * hds * lq * lk * 64)
            native.check(native.lib().dbir_set_option(1, 2), "dbir_set_option")
    if want("norm"):
        for (b, hw, c) in [(B, 4096, 320), (B, 4096, 960), (B, 1024, 640), (B, 256, 1280), (B, 64, 2560),
                           (8, 262144, 128)]:
            x = rnd(b, hw, c)
            g, be = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
            y = torch.empty_like(x)
            sec = timeit(lambda: ops.groupnorm(x, g, be, 1e-5, True, out=y))
            report(f"groupnorm+silu B{b} HW{hw} C{c}", "3-launch", sec, bytes_=2.0 * x.numel() * 2)
        for (rows_, c) in [(65536, 320), (16384, 640), (4096, 1280)]:
            x = rnd(rows_, c)
            g, be = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
            y = torch.empty_like(x)
            sec = timeit(lambda: ops.layernorm(x, g, be, out=y))
            report(f"layernorm {rows_}x{c}", "wave/row", sec, bytes_=2.0 * x.numel() * 2)
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
