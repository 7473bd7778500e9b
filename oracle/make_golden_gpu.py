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


# ---- big-tensor-safe primitives (C5 only) ------------------------------------------------------------------------------------
# The untiled VAE decoder at 4096 x 4096 holds activations of 2^32 elements (256 channels x 4096 x 4096); MIOpen / ATen
# kernels index with 32 bits and fail there ("invalid configuration argument", call 2 of round 5).  The wrappers below evaluate
# the SAME operators in pieces of at most BIG_LIMIT elements: convolutions per block of output channels as a sum over blocks of
# input channels (linear in the input; f32 partial sums added in channel order), GroupNorm per group (a group's statistics
# involve only its own channels: identical math), swish / nearest interpolation per channel block (elementwise).
BIG_LIMIT = 1 << 30


def _chunks(n: int, per: int):
    per = max(1, per)
    return [(i, min(n, i + per)) for i in range(0, n, per)]


def big_conv(w, x, stride=1, padding=0):
    import torch.nn.functional as F
    wt = w["weight"]
    b = w["bias"] if w.has("bias") else None
    px = x.numel() // max(x.shape[1], 1)                                   # elements per channel (batch included)
    if x.numel() <= BIG_LIMIT and px // (stride * stride) * wt.shape[0] <= BIG_LIMIT:
        return F.conv2d(x, wt, b, stride=stride, padding=padding)
    outs = []
    for o0, o1 in _chunks(wt.shape[0], BIG_LIMIT // max(px // (stride * stride), 1)):
        acc = None
        for i0, i1 in _chunks(wt.shape[1], BIG_LIMIT // px):
            y = F.conv2d(x[:, i0:i1].contiguous(), wt[o0:o1, i0:i1].contiguous(), None, stride=stride, padding=padding)
            acc = y if acc is None else acc.add_(y)
        if b is not None:
            acc.add_(b[o0:o1].view(1, -1, 1, 1))
        outs.append(acc)
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)


def big_gnorm(w, x, eps):
    import torch.nn.functional as F
    if x.numel() <= BIG_LIMIT:
        return F.group_norm(x, 32, w["weight"], w["bias"], eps)
    cpg = x.shape[1] // 32
    out = torch.empty_like(x)
    for g in range(32):
        sl = slice(g * cpg, (g + 1) * cpg)
        out[:, sl] = F.group_norm(x[:, sl].contiguous(), 1, w["weight"][sl], w["bias"][sl], eps)
    return out


def big_swish(x):
    if x.numel() <= BIG_LIMIT:
        return x * torch.sigmoid(x)
    out = torch.empty_like(x)
    for c0, c1 in _chunks(x.shape[1], BIG_LIMIT // (x.numel() // x.shape[1])):
        out[:, c0:c1] = x[:, c0:c1] * torch.sigmoid(x[:, c0:c1])
    return out


class _BigF:
    """torch.nn.functional with a channel-blocked nearest `interpolate` (everything else passes through)."""

    def __init__(self, F):
        self._F = F

    def __getattr__(self, name):
        return getattr(self._F, name)

    def interpolate(self, x, *a, **k):
        sf = k.get("scale_factor") or (a[1] if len(a) > 1 else None)
        if k.get("mode") != "nearest" or sf is None or x.dim() != 4 or x.numel() * int(sf) ** 2 <= BIG_LIMIT:
            return self._F.interpolate(x, *a, **k)
        per = BIG_LIMIT // (x.numel() // x.shape[1] * int(sf) ** 2)
        return torch.cat([self._F.interpolate(x[:, c0:c1].contiguous(), *a, **k) for c0, c1 in _chunks(x.shape[1], per)], dim=1)


class big_tensors:
    """with big_tensors(): oracle.nets runs on the wrappers above."""

    def __enter__(self):
        self.saved = (nets.conv, nets.gnorm, nets._swish, nets.F, nets.vae_attn)
        nets.conv, nets.gnorm, nets._swish, nets.F, nets.vae_attn = big_conv, big_gnorm, big_swish, _BigF(nets.F), vae_attn_chunked
        return self

    def __exit__(self, *exc):
        nets.conv, nets.gnorm, nets._swish, nets.F, nets.vae_attn = self.saved
        return False


def build(device):
    from diffbir_amd import configs
    cldm_cfg, swin_cfg = cases.get_cfgs("full")
    W = cases.synth_weights(cldm_cfg, swin_cfg, 0)
    gm = np.load(os.path.join(GOLD, "full_modules.npz"))
    table = {"": torch.tensor(gm["tokens"][0]), cases.NEG_PROMPT: torch.tensor(gm["tokens"][1])}
    return GpuOraclePipeline(W, cldm_cfg, swin_cfg, configs.get("DIFFUSION_V21"),
                             lambda txts: torch.stack([table[t] for t in txts]), device)


def finish_from_latent(orc, lqspec, z):
    """The tail of OraclePipeline.run / apply_cldm (reference pipeline.py:218-231, 306-320) from a saved latent: stage-1 output
    again (deterministic), VAE decode, wavelet colour fix, uint8."""
    import torch.nn.functional as F
    from .pipeline import wavelet_reconstruction
    lq = cases.make_lq(*lqspec)
    x = torch.tensor(lq, dtype=torch.float32).div(255).clamp(0, 1).permute(0, 3, 1, 2).contiguous()
    cond_img = orc.apply_cleaner(x)
    h0, w0 = cond_img.shape[2:]
    sample = orc.vae_decode(z)[:, :, :h0, :w0]
    sample = F.interpolate(wavelet_reconstruction((sample + 1) / 2, cond_img), size=tuple(x.shape[2:]), mode="bicubic",
                           antialias=True)
    return (sample * 255.0).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()


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
        zpath = os.path.join(ROOT, "oracle", "_c5_latent.npz")   # (git-ignored) a run that died after the sampling loop resumes here
        with big_tensors():
            if os.path.exists(zpath):
                out, dt = finish_from_latent(orc, lqspec, torch.tensor(np.load(zpath)["z"])), float(np.load(zpath)["seconds"])
            else:
                taps = {}
                t0 = time.time()
                try:
                    out = orc.run(cases.make_lq(*lqspec), steps, neg_prompt=cases.NEG_PROMPT, cfg_scale=4.0, sampler_type=sampler,
                                  randn=cases.NoiseStream(seed), taps=taps, **kw)
                except Exception:
                    if "z" in taps:   # ~20 PFLOP of sampling are done: keep the latent so that only the decode is repeated
                        np.savez_compressed(os.path.join(OUT, "c5_latent.npz"), z=taps["z"].float().cpu().numpy(),
                                            seconds=np.float64(time.time() - t0))
                        print("sampling finished, decode failed: latent saved to gpurun_out/golden_gpu/c5_latent.npz "
                              "(copy it to oracle/_c5_latent.npz and re-run `c5`)", flush=True)
                    raise
                dt = time.time() - t0
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
