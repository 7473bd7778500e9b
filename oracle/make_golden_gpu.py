"""ORACLE — TEST INFRASTRUCTURE.  Goldens at the BENCHMARKED shapes of BASELINE configs C3 / C4.

The unmodified reference on CPU cannot produce them (1.8 h for C4 at 10 of its 50 steps), so the ORACLE (oracle/pipeline.py,
the fp32 restatement of the reference that tests/test_oracle_golden.py pins to the reference's own outputs) is run with its
network evaluations on the GPU in **fp32 on plain PyTorch-ROCm** (no engine kernel, no reduced precision):

    gpurun -- 'python -m oracle.make_golden_gpu'         # writes gpurun_out/golden_gpu/*.npz + report.json
    cp gpurun_out/golden_gpu/full_*.npz tests/golden/ && python tools/golden_manifest.py

What runs where: the host-side sampling math (schedules, CFG mix, x0 / posterior, DPM-Solver++ updates, the tile blend in
the reference's sequential order, noise from the CPU generator the reference consumes) stays on the CPU exactly as in the
CPU oracle; only SwinIR, the VAE, the CLIP tower and ControlNet + UNet evaluations are moved to the device
(`GpuOraclePipeline`).  The tiles of one tiled evaluation are evaluated as device batches (per-sample math: GroupNorm /
LayerNorm / attention never mix samples) and blended on the host tile by tile.

The chain reference -> oracle (CPU) -> oracle (GPU fp32) is CLOSED before anything is written: the GPU oracle must reproduce
the committed CPU-REFERENCE goldens of C2 (batch 2, 50 steps) and of C4 (2048 x 2048, 10 steps) — the report records PSNR and
the largest uint8 difference (fp32 summation order on another device: a few +-1 LSB pixels are expected, nothing more).

The product never imports this module (nor anything under oracle/).
"""
import json
import os
import sys
import time

import numpy as np
import torch

from . import cases, nets, sampling
from .pipeline import OraclePipeline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "golden_gpu")
GOLD = os.path.join(ROOT, "tests", "golden")

# name -> (lq spec, steps, sampler, seed, kwargs) — tests/test_pipeline_gpu.py GPU_ORACLE_CASES mirrors this table
CASES = {
    "c3_dpm20_b4": ((27, 4, 512, 512), 20, "dpm++_m2", 231, {}),                      # C3 at the bench's batch 4 per GPU
    "c4_tiled2048_spaced50": ((25, 1, 2048, 2048), 50, "spaced", 231,                 # C4 exactly as benchmarked: 49 tiles x 50 steps
                              dict(cldm_tiled=True, cldm_tile_size=512, cldm_tile_stride=256)),
}
# committed CPU-reference goldens the GPU oracle must reproduce first: file -> (lq spec, steps, sampler, seed, kwargs)
CHAIN = {
    "full_c2_spaced50_b2.npz": ((21, 2, 512, 512), 50, "spaced", 231, {}),
    "full_c4_tiled2048_spaced10.npz": ((25, 1, 2048, 2048), 10, "spaced", 231,
                                       dict(cldm_tiled=True, cldm_tile_size=512, cldm_tile_stride=256)),
}
# BASELINE config C5 as benchmarked, ONE of its four images: 4096 x 4096, 225 tiles per evaluation, all 50 spaced steps
# (22 500 fp32 tile evaluations, ~20 PFLOP: about five minutes of device time) — the 15 x 15 tile schedule, the 512 x 512
# latent blend and the untiled VAE at 4096 x 4096 (262 144-token mid-block attention, evaluated exactly in query chunks).
# Stored as a 4x-strided subsample + eight full-resolution 256 x 256 crops (the uint8 output is 50 MB).
C5_CASE = ("c5_tiled4096_spaced50", (28, 1, 4096, 4096), 50, "spaced", 231,
           dict(cldm_tiled=True, cldm_tile_size=512, cldm_tile_stride=256))
C5_CROPS = [(0, 0), (0, 3840), (3840, 0), (3840, 3840), (1920, 1920), (1000, 2500), (2500, 1000), (3000, 3000)]
TILE_CHUNK = 25   # tiles per device batch (x batch samples)


def c5_pack(out: np.ndarray) -> dict:
    """uint8 [1, 4096, 4096, 3] -> the stored views (tests/test_pipeline_gpu.py compares the same views of the engine output)."""
    return dict(strided=np.ascontiguousarray(out[:, ::4, ::4]),
                crops=np.stack([out[0, y:y + 256, x:x + 256] for y, x in C5_CROPS]))


def vae_attn_chunked(w, x, chunk: int = 8192):
    """nets.vae_attn (reference vae.py:253-282) with the softmax(QK^T)V evaluated exactly per block of `chunk` queries — the
    same values, without the L x L matrix (L = 262 144 at 4096 x 4096)."""
    import torch.nn.functional as F
    b, c, hh, ww = x.shape
    h = nets.gnorm(w("norm"), x, 1e-6)
    q, k, v = (nets.conv(w(n), h).reshape(b, c, hh * ww).permute(0, 2, 1) for n in ("q", "k", "v"))
    o = torch.cat([F.scaled_dot_product_attention(q[:, None, i:i + chunk], k[:, None], v[:, None])[:, 0]
                   for i in range(0, hh * ww, chunk)], dim=1)
    o = o.permute(0, 2, 1).reshape(b, c, hh, ww)
    return x + nets.conv(w("proj_out"), o)


class GpuOraclePipeline(OraclePipeline):
    """OraclePipeline whose four network entry points run on `device` in fp32; everything else is the CPU oracle's code."""

    def __init__(self, W, cldm_cfg, swinir_cfg, diffusion_cfg, tokenize, device):
        super().__init__(W, cldm_cfg, swinir_cfg, diffusion_cfg, tokenize)
        self.dev = torch.device(device)
        self.W = {m: {k: v.to(self.dev) for k, v in sd.items()} for m, sd in W.items()}

    def model(self, x, t, cond):
        d = self.dev
        out = nets.cldm_forward(self.W, self.cldm_cfg, x.to(d), t.to(d), cond["c_txt"].to(d), cond["c_img"].to(d),
                                self.control_scales)
        return out.float().cpu()

    def prepare_condition(self, img, txt):
        d = self.dev
        return dict(
            c_txt=nets.clip_text_encode(self.W["clip"], self.cldm_cfg["clip_cfg"], self.tokenize(txt).to(d)).cpu(),
            c_img=nets.vae_encode_mode(self.W["vae"], self.cldm_cfg["vae_cfg"], (img * 2 - 1).to(d), self.scale_factor).cpu())

    def vae_decode(self, z):
        return nets.vae_decode(self.W["vae"], self.cldm_cfg["vae_cfg"], z.to(self.dev), self.scale_factor).cpu()

    def cleaner(self, x):
        return nets.swinir_forward(self.W["swinir"], self.swinir_cfg, x.to(self.dev)).cpu()


def batched_tile_model(model, size: int, stride: int):
    """sampling.tile_model (reference spaced_sampler.py:204-219 over utils/common.py:172-232) with the tiles of one evaluation
    sent to the device as batches; the weighted blend runs on the host in the reference's tile order with the same f32
    operations (out += eps_tile * w; count += w; out / count)."""

    def tiled(x, t, cond):
        b, c, h, w = x.shape
        wins = sampling.sliding_windows(h, w, size, stride)
        wt = torch.tensor(sampling.gaussian_weights(size, size)[None, None], dtype=x.dtype)
        out = torch.zeros((b, c, h, w), dtype=x.dtype)
        count = torch.zeros_like(out, dtype=torch.float32)
        for i in range(0, len(wins), TILE_CHUNK):
            ws = wins[i:i + TILE_CHUNK]
            xt = torch.cat([x[..., hi:he, wi:we] for hi, he, wi, we in ws])
            ci = torch.cat([cond["c_img"][..., hi:he, wi:we] for hi, he, wi, we in ws])
            eps = model(xt, t.repeat(len(ws)), {"c_txt": cond["c_txt"].repeat(len(ws), 1, 1), "c_img": ci})
            for k, (hi, he, wi, we) in enumerate(ws):
                out[..., hi:he, wi:we] += eps[k * b:(k + 1) * b] * wt
                count[..., hi:he, wi:we] += wt
        return out / count

    return tiled


def build(device):
    from diffbir_amd import configs
    cldm_cfg, swin_cfg = cases.get_cfgs("full")
    W = cases.synth_weights(cldm_cfg, swin_cfg, 0)
    gm = np.load(os.path.join(GOLD, "full_modules.npz"))
    table = {"": torch.tensor(gm["tokens"][0]), cases.NEG_PROMPT: torch.tensor(gm["tokens"][1])}
    return GpuOraclePipeline(W, cldm_cfg, swin_cfg, configs.get("DIFFUSION_V21"),
                             lambda txts: torch.stack([table[t] for t in txts]), device)


def run_case(orc, spec):
    lqspec, steps, sampler, seed, kw = spec
    torch.cuda.synchronize()
    t0 = time.time()
    out = orc.run(cases.make_lq(*lqspec), steps, neg_prompt=cases.NEG_PROMPT, cfg_scale=4.0, sampler_type=sampler,
                  randn=cases.NoiseStream(seed), **kw)
    torch.cuda.synchronize()
    return out, time.time() - t0


@torch.no_grad()
def main(argv):
    assert torch.cuda.is_available(), "oracle.make_golden_gpu needs a GPU (fp32 PyTorch-ROCm)"
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    os.makedirs(OUT, exist_ok=True)
    sampling.tile_model = batched_tile_model
    orc = build("cuda:0")
    report = dict(device=torch.cuda.get_device_name(0), torch=torch.__version__, precision="fp32 (TF32 off)", chain={}, cases={})
    only = set(argv)
    ok = True
    for fname, spec in CHAIN.items():
        if only and "chain" not in only and fname not in only:
            continue
        if only == {"c5"}:
            continue
        ref = np.load(os.path.join(GOLD, fname))["out"]
        out, dt = run_case(orc, spec)
        diff = np.abs(out.astype(np.int16) - ref.astype(np.int16))
        rec = dict(psnr_db=cases.psnr_u8(out, ref), max_abs_u8=int(diff.max()), frac_pixels_differ=float((diff > 0).mean()),
                   seconds=dt, what="GPU-fp32 oracle vs the committed CPU-reference golden")
        report["chain"][fname] = rec
        print("chain", fname, rec, flush=True)
        # bar: the two fp32 evaluations agree to rounding — >= 60 dB and no pixel off by more than 2 LSB
        if not (rec["psnr_db"] >= 60.0 and rec["max_abs_u8"] <= 2):
            ok = False
    report["chain_closed"] = ok
    if ok:
        for name, spec in CASES.items():
            if only and name not in only and "cases" not in only:
                continue
            out, dt = run_case(orc, spec)
            np.savez_compressed(os.path.join(OUT, f"full_{name}.npz"), out=out, oracle_gpu_seconds=np.float64(dt),
                                oracle="oracle.make_golden_gpu (fp32, PyTorch-ROCm): " + report["device"])
            report["cases"][name] = dict(shape=list(out.shape), seconds=dt)
            print("case", name, out.shape, f"{dt:.1f} s", flush=True)
    if "c5" in only:   # on request only (about five minutes of device time); the chain above is checked by the default run
        name, lqspec, steps, sampler, seed, kw = C5_CASE
        orig = nets.vae_attn
        nets.vae_attn = vae_attn_chunked
        try:
            out, dt = run_case(orc, (lqspec, steps, sampler, seed, kw))
        finally:
            nets.vae_attn = orig
        np.savez_compressed(os.path.join(OUT, f"full_{name}.npz"), oracle_gpu_seconds=np.float64(dt),
                            oracle="oracle.make_golden_gpu (fp32, PyTorch-ROCm): " + report["device"], **c5_pack(out))
        report["cases"][name] = dict(shape=list(out.shape), seconds=dt)
        print("case", name, out.shape, f"{dt:.1f} s", flush=True)
    with open(os.path.join(OUT, "report.json" if "c5" not in only else "report_c5.json"), "w") as f:
        json.dump(report, f, indent=1)
    if not ok:
        print("CHAIN NOT CLOSED: the GPU oracle does not reproduce the CPU-reference goldens; nothing written", flush=True)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
