#!/usr/bin/env python
"""How much does taking GroupNorm statistics from the GEMM epilogue (ops.GnPartials) move the result?  Tiny and full
configurations, batch 2 vs 2 x batch 1, with the switch on / off (PSNR between u8 outputs; same noise everywhere)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffbir_amd.model import unet  # noqa: E402
from oracle import cases  # noqa: E402
from tests.helpers import build_engine  # noqa: E402


def run(pipe, lq, draws, sl):
    it = iter([d[sl] for d in draws])
    pipe.randn = lambda shape: next(it)
    args = (3, 1.0, False, 512, 256, False, 256, False, 256, False, 512, 256, "", cases.NEG_PROMPT, 4.0, "noise",
            "spaced", 0, False, 0, 0, 300, 1, 1, 1)
    return pipe.run(lq[sl], *args)


def main():
    dev = torch.device("cuda:0")
    for cfg in sys.argv[1:] or ["tiny"]:
        pipe, cldm, swin = build_engine(cfg, "DIFFUSION_V21", dev, torch.float16)
        lq = cases.make_lq(5, 2, 512, 512)
        full = cases.NoiseStream(11)
        draws = []

        def rec(shape):
            t = full(shape)
            draws.append(t)
            return t
        pipe.randn = rec
        args = (3, 1.0, False, 512, 256, False, 256, False, 256, False, 512, 256, "", cases.NEG_PROMPT, 4.0, "noise",
                "spaced", 0, False, 0, 0, 300, 1, 1, 1)
        pipe.run(lq, *args)
        res = {}
        for on in (False, True):
            unet.GN_EPI_STATS = on
            cldm.reset_graphs()
            res[on, "b2"] = run(pipe, lq, draws, slice(0, 2))
            res[on, "b1"] = [run(pipe, lq, draws, slice(i, i + 1)) for i in range(2)]
        p = cases.psnr_u8
        unet.GN_EPI_STATS = False
        cldm.reset_graphs()
        rep2 = run(pipe, lq, draws, slice(0, 2))
        rep1 = run(pipe, lq, draws, slice(0, 1))
        print(cfg, "same settings run twice (stats off): batch 2 bit-identical:", bool((rep2 == res[False, "b2"]).all()),
              " batch 1 bit-identical:", bool((rep1 == res[False, "b1"][0]).all()),
              " differing u8 values:", int((rep2 != res[False, "b2"]).sum()), int((rep1 != res[False, "b1"][0]).sum()))
        for on in (False, True):
            print(cfg, f"epilogue stats {'on ' if on else 'off'}: batch 2 vs batch 1:",
                  [round(p(res[on, 'b1'][i], res[on, 'b2'][i:i + 1]), 2) for i in range(2)])
        print(cfg, "batch 2: on vs off:", round(p(res[True, "b2"], res[False, "b2"]), 2),
              " batch 1: on vs off:", [round(p(res[True, "b1"][i], res[False, "b1"][i]), 2) for i in range(2)])


if __name__ == "__main__":
    main()
