"""ORACLE — TEST INFRASTRUCTURE. Generates tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference, CPU fp32) on seeded synthetic weights/inputs.  Run here (the GPU box has no reference):

    python -m oracle.make_golden tiny            # seconds..minutes
    python -m oracle.make_golden tiny_options    # option paths of Pipeline.run on the tiny config
    python -m oracle.make_golden host_tables     # schedules / tile windows / blend weights from the reference's functions
    python -m oracle.make_golden cleaners        # BSRNet / SCUNet modules (tiny + shipped configs) and their pipelines (tiny)
    python -m oracle.make_golden full_modules    # full-size nets, module level (a few minutes)
    python -m oracle.make_golden full_pipeline   # 1x512x512, 50 spaced steps + CFG (≈10 min on 8 cores)
    python -m oracle.make_golden full_configs    # BASELINE configs at full size: C2 (batch 2, spaced 50), C3 (batch 2,
                                                 # dpm++_m2 20 steps), C4 (1x1024x1024 tiled 512/256, spaced 10)  ≈45 min

Fixtures hold only outputs + token ids (+ tiny inputs); weights/inputs are re-derived from seeds by
oracle/cases.py on both sides.
"""
import os
import sys
import time

import numpy as np
import torch

from . import cases
from .ref_import import load_reference

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def build_reference(R, cfg_name: str, diffusion: dict, seed: int = 0):
    cldm_cfg, swin_cfg = cases.get_cfgs(cfg_name)
    W = cases.synth_weights(cldm_cfg, swin_cfg, seed)
    with cases.quiet():
        cldm = R.ControlLDM(**cldm_cfg).eval()
        swin = R.SwinIR(**swin_cfg).eval()
    cldm.unet.load_state_dict(W["unet"], strict=True)
    cldm.controlnet.load_state_dict(W["controlnet"], strict=True)
    cldm.vae.load_state_dict(W["vae"], strict=True)
    cldm.clip.load_state_dict(W["clip"], strict=True)
    missing, unexpected = swin.load_state_dict(W["swinir"], strict=False)
    assert not unexpected and all(k.endswith(("attn_mask", "relative_position_index")) for k in missing), missing
    diff = R.Diffusion(**diffusion)
    return cldm, swin, diff, W


def tokens_of(R, prompts):
    from diffbir.model.open_clip import tokenize
    return tokenize(list(prompts))


def run_pipeline(R, cldm, swin, diff, lq, steps, sampler, seed, cfg=4.0, tiled=False, tile=512, stride=256,
                 cleaner_tiled=False, strength=1.0, start="noise", noise_aug=0, rescale_cfg=False, pos=""):
    pipe = R.SwinIRPipeline(swin, cldm, diff, None, "cpu")
    torch.manual_seed(seed)
    with cases.quiet():
        return pipe.run(lq, steps, strength, cleaner_tiled, 512, 256, False, 256, False, 256, tiled, tile, stride,
                        pos, cases.NEG_PROMPT, cfg, start, sampler, noise_aug, rescale_cfg, 0, 0, 300, 1, 1, 1)


# option paths of Pipeline.run / apply_cldm (pipeline.py:146-174, 371-397; sampler.py:31-38): name -> (lq spec, kwargs)
OPTION_CASES = {
    "cond_start": ((3, 1, 512, 512), dict(steps=4, sampler="spaced", seed=7, start="cond")),
    "noise_aug": ((3, 1, 512, 512), dict(steps=4, sampler="spaced", seed=7, noise_aug=120)),
    "rescale_cfg": ((3, 1, 512, 512), dict(steps=4, sampler="spaced", seed=7, rescale_cfg=True, cfg=3.0)),
    "cfg1": ((3, 1, 512, 512), dict(steps=4, sampler="spaced", seed=7, cfg=1.0)),
    "strength": ((3, 1, 512, 512), dict(steps=4, sampler="dpm++_m2", seed=7, strength=0.6)),
    "cleaner_tiled": ((9, 1, 600, 712), dict(steps=3, sampler="spaced", seed=5, cleaner_tiled=True)),
    "small_upsized": ((13, 1, 300, 256), dict(steps=3, sampler="spaced", seed=5)),
}


@torch.no_grad()
def gen_cleaners(R):
    """BSRNet / SCUNet: module outputs of the reference's RRDBNet / SCUNet (tiny + shipped configs) and
    end-to-end BSRNetPipeline / SCUNetPipeline runs on the tiny ControlLDM.  Also asserts oracle == reference (exact)."""
    from diffbir_amd import configs
    from diffbir.model.bsrnet import RRDBNet
    from diffbir.model.scunet import SCUNet
    from diffbir.pipeline import BSRNetPipeline, SCUNetPipeline
    from . import nets
    g, mods = {}, {}
    for name in cases.CLEANERS:
        cfg, W, x = cases.cleaner_case(name)
        with cases.quiet():
            m = (RRDBNet if name.startswith("bsrnet") else SCUNet)(**cfg).eval()
        m.load_state_dict(W, strict=True)
        y = m(x)
        o = (nets.rrdbnet_forward if name.startswith("bsrnet") else nets.scunet_forward)(W, cfg, x)
        assert torch.equal(o, y) or (o - y).abs().max() < 1e-5 * y.abs().max(), name
        g[name] = y.numpy()
        mods[name] = m
        print(name, y.shape, float(y.abs().max()))
    cldm, _, diff, _ = build_reference(R, "tiny", configs.get("DIFFUSION_V21"))
    for name, (cleaner, lqspec, kw) in cases.CLEANER_PIPELINES.items():
        if cleaner.startswith("bsrnet"):
            pipe = BSRNetPipeline(mods[cleaner], cldm, diff, None, "cpu", kw["upscale"])
        else:
            pipe = SCUNetPipeline(mods[cleaner], cldm, diff, None, "cpu")
        torch.manual_seed(kw["seed"])
        with cases.quiet():
            g["pipe_" + name] = pipe.run(
                cases.make_lq(*lqspec), kw["steps"], 1.0, kw.get("cleaner_tiled", False), kw.get("cleaner_tile", 512),
                kw.get("cleaner_stride", 256), False, 256, False, 256, False, 512, 256, "", cases.NEG_PROMPT, 4.0, "noise",
                "spaced", 0, False, 0, 0, 300, 1, 1, 1)
        print(name, g["pipe_" + name].shape)
    np.savez_compressed(os.path.join(OUT, "cleaners.npz"), **g)


@torch.no_grad()
def gen_tiny_options(R):
    from diffbir_amd import configs
    cldm, swin, diff, W = build_reference(R, "tiny", configs.get("DIFFUSION_V21"))
    g = {}
    for name, (lqspec, kw) in OPTION_CASES.items():
        kw = dict(kw)
        g[name] = run_pipeline(R, cldm, swin, diff, cases.make_lq(*lqspec), kw.pop("steps"), kw.pop("sampler"),
                               kw.pop("seed"), **kw)
        print(name, g[name].shape)
    np.savez_compressed(os.path.join(OUT, "tiny_options.npz"), **g)


@torch.no_grad()
def gen_modules(R, cfg_name: str, tag: str, img: int, diffusion):
    cldm, swin, diff, W = build_reference(R, cfg_name, diffusion)
    g = {}
    rs = cases.NoiseStream(7)
    x = torch.tensor(cases.make_lq(11, 2, img, img)).float().div(255).permute(0, 3, 1, 2).contiguous()
    g["swinir_out"] = swin(x).numpy()
    g["vae_mode"] = cldm.vae_encode(x * 2 - 1, sample=False).numpy()
    z = rs((2, 4, img // 8, img // 8))
    g["vae_dec"] = cldm.vae_decode(z).numpy()
    toks = tokens_of(R, ["", cases.NEG_PROMPT])
    g["tokens"] = toks.numpy()
    c_txt = cldm.clip(toks)
    g["c_txt"] = c_txt.numpy()
    xn = rs((2, 4, img // 8, img // 8))
    c_img = rs((2, 4, img // 8, img // 8)) * 0.5
    cldm.control_scales = [0.9] * 13
    t_int = torch.tensor([999, 381], dtype=torch.long)
    g["eps_int_t"] = cldm(xn, t_int, dict(c_txt=c_txt, c_img=c_img)).numpy()
    t_f = torch.tensor([949.0365, 49.95], dtype=torch.float32)
    g["eps_float_t"] = cldm(xn, t_f, dict(c_txt=c_txt, c_img=c_img)).numpy()
    ctrl = cldm.controlnet(x=xn, hint=c_img, timesteps=t_int, context=c_txt)
    g["control_0"] = ctrl[0].numpy()
    g["control_12"] = ctrl[12].numpy()
    np.savez_compressed(os.path.join(OUT, f"{tag}_modules.npz"), **g)
    print(tag, "modules done", {k: v.shape for k, v in g.items()})


@torch.no_grad()
def gen_tiny_pipelines(R):
    g = {}
    for ver, dcfg in (("v21", "DIFFUSION_V21"), ("v2", "DIFFUSION_V2")):
        from diffbir_amd import configs
        cldm, swin, diff, W = build_reference(R, "tiny", configs.get(dcfg))
        lq = cases.make_lq(3, 1, 512, 512)
        g[f"spaced6_{ver}"] = run_pipeline(R, cldm, swin, diff, lq, 6, "spaced", 231)
        if ver == "v21":
            g["dpm10_v21"] = run_pipeline(R, cldm, swin, diff, lq, 10, "dpm++_m2", 231)
            lq2 = cases.make_lq(5, 2, 512, 512)
            g["spaced4_b2_v21"] = run_pipeline(R, cldm, swin, diff, lq2, 4, "spaced", 99)
            lq3 = cases.make_lq(9, 1, 600, 712)  # not multiples of 64/8: pad paths + edge-flush tile windows
            g["spaced3_pad_v21"] = run_pipeline(R, cldm, swin, diff, lq3, 3, "spaced", 5)
            g["spaced3_tiled_v21"] = run_pipeline(R, cldm, swin, diff, lq3, 3, "spaced", 5, tiled=True)
            g["dpm10_tiled_v21"] = run_pipeline(R, cldm, swin, diff, lq3, 10, "dpm++_m2", 5, tiled=True)
        else:
            g["dpm10_v2"] = run_pipeline(R, cldm, swin, diff, lq, 10, "dpm++_m2", 231)
    np.savez_compressed(os.path.join(OUT, "tiny_pipeline.npz"), **g)
    print("tiny pipelines done", {k: v.shape for k, v in g.items()})


@torch.no_grad()
def gen_full_pipeline(R):
    from diffbir_amd import configs
    cldm, swin, diff, W = build_reference(R, "full", configs.get("DIFFUSION_V21"))
    lq = cases.make_lq(3, 1, 512, 512)
    t0 = time.time()
    out = run_pipeline(R, cldm, swin, diff, lq, 50, "spaced", 231)
    dt = time.time() - t0
    np.savez_compressed(os.path.join(OUT, "full_pipeline.npz"), spaced50_v21=out,
                        ref_cpu_seconds=np.float64(dt), ref_cpu_threads=np.int64(torch.get_num_threads()))
    print("full pipeline done in", dt, "s")


# BASELINE configs at FULL network size: name -> (lq spec, steps, sampler, seed, kwargs)
FULL_CONFIG_CASES = {
    "c2_spaced50_b2": ((21, 2, 512, 512), 50, "spaced", 231, {}),
    "c3_dpm20_b2": ((22, 2, 512, 512), 20, "dpm++_m2", 231, {}),
    "c4_tiled1024_spaced10": ((23, 1, 1024, 1024), 10, "spaced", 231, dict(tiled=True, tile=512, stride=256)),
    # round 3: the BENCHMARKED shapes themselves: C2 at bench batch 8, C4 at 2048x2048 / 49 tiles
    "c2_spaced50_b8": ((24, 8, 512, 512), 50, "spaced", 231, {}),
    "c4_tiled2048_spaced10": ((25, 1, 2048, 2048), 10, "spaced", 231, dict(tiled=True, tile=512, stride=256)),
}


@torch.no_grad()
def gen_full_configs(R, only=None):
    """One npz per case (each takes minutes to an hour of CPU): tests/golden/full_<name>.npz with the reference's uint8
    output, its wall time and thread count."""
    from diffbir_amd import configs
    cldm, swin, diff, W = build_reference(R, "full", configs.get("DIFFUSION_V21"))
    for name, (lqspec, steps, sampler, seed, kw) in FULL_CONFIG_CASES.items():
        if only and name not in only:
            continue
        t0 = time.time()
        out = run_pipeline(R, cldm, swin, diff, cases.make_lq(*lqspec), steps, sampler, seed, **kw)
        dt = time.time() - t0
        np.savez_compressed(os.path.join(OUT, f"full_{name}.npz"), out=out, ref_cpu_seconds=np.float64(dt),
                            ref_cpu_threads=np.int64(torch.get_num_threads()))
        print(name, out.shape, f"{dt:.1f} s", flush=True)


@torch.no_grad()
def gen_full_bf16(R):
    """BASELINE config C5's precision at FULL network size: one tiled case (768x768, tile 512 /
    stride 256 = 4 tiles, 3 spaced steps + CFG) — the reference's fp32 output (the golden) and the PSNR of the reference's
    OWN bf16 run (cast_dtype + autocast, what `--precision bf16` selects) against it: the yardstick the engine's bf16
    result is held to (within 1.5 dB, tests/test_pipeline_gpu.py)."""
    from diffbir_amd import configs
    lq = cases.make_lq(26, 1, 768, 768)
    kw = dict(tiled=True, tile=512, stride=256)
    cldm, swin, diff, W = build_reference(R, "full", configs.get("DIFFUSION_V21"))
    t0 = time.time()
    ref = run_pipeline(R, cldm, swin, diff, lq, 3, "spaced", 231, **kw)
    t_f32 = time.time() - t0
    print("fp32", ref.shape, f"{t_f32:.1f} s", flush=True)
    cldm.cast_dtype(torch.bfloat16)
    t0 = time.time()
    with torch.autocast("cpu", torch.bfloat16):
        low = run_pipeline(R, cldm, swin, diff, lq, 3, "spaced", 231, **kw)
    psnr = cases.psnr_u8(low, ref)
    print("bf16", f"{time.time() - t0:.1f} s  PSNR vs fp32 {psnr:.2f} dB", flush=True)
    np.savez_compressed(os.path.join(OUT, "full_c5_tiled768_spaced3_bf16.npz"), out=ref, ref_bf16_psnr=np.float64(psnr),
                        ref_cpu_seconds=np.float64(t_f32), ref_cpu_threads=np.int64(torch.get_num_threads()))


# round 4: full-size fixtures for the paths that were pinned on the tiny config only
FULL_SAMPLER_CASES = {   # name -> (sampler, steps): 1 x 512 x 512, CFG 4, seed 17 (as gen_samplers)
    "ddim5": ("ddim", 5),
    "edm_dpm++_2m_6": ("edm_dpm++_2m", 6),
}


@torch.no_grad()
def gen_full_extra(R):
    """tests/golden/full_extra.npz: (a) the reference's VAEHook at FULL network size — encode of a 1024 x 1024 image with
    encoder tile 512, decode of a 128 x 128 latent with decoder tile 64 (what bench.py --vae-tiled switches on for every
    N > 1 tiled run); (b) DDIM and edm_dpm++_2m pipelines at full size."""
    from diffbir_amd import configs
    cldm, swin, diff, W = build_reference(R, "full", configs.get("DIFFUSION_V21"))
    g = {}
    x = torch.tensor(cases.make_lq(51, 1, 1024, 1024)).float().div(255).permute(0, 3, 1, 2).contiguous()
    t0 = time.time()
    with cases.quiet():
        g["enc_tiled_512"] = cldm.vae_encode(x * 2 - 1, sample=False, tiled=True, tile_size=512).numpy()
        print("enc_tiled_512", g["enc_tiled_512"].shape, f"{time.time() - t0:.1f} s", flush=True)
        z = cases.NoiseStream(52)((1, 4, 128, 128))
        g["dec_tiled_64"] = cldm.vae_decode(z, tiled=True, tile_size=64).numpy().astype(np.float16)   # 12 MB as f32: stored f16
        print("dec_tiled_64", g["dec_tiled_64"].shape, f"{time.time() - t0:.1f} s", flush=True)
    pipe = R.SwinIRPipeline(swin, cldm, diff, None, "cpu")
    for name, (sampler, steps) in FULL_SAMPLER_CASES.items():
        torch.manual_seed(17)
        with cases.quiet():
            g[name] = pipe.run(cases.make_lq(53, 1, 512, 512), steps, 1.0, False, 512, 256, False, 256, False, 256, False,
                               512, 256, "", cases.NEG_PROMPT, 4.0, "noise", sampler, 0, False, 0, 0, 300, 1, 1, 1)
        print(name, g[name].shape, f"{time.time() - t0:.1f} s", flush=True)
    np.savez_compressed(os.path.join(OUT, "full_extra.npz"), **g)


@torch.no_grad()
def gen_tiled_vae(R):
    """Tiled VAE: the reference's VAEHook through ControlLDM.vae_encode / vae_decode with `tiled=True`
    (cldm.py:99-111,127-138) on the tiny config, edge tiles included, + its tile geometry for several sizes."""
    import importlib
    import json
    from diffbir_amd import configs
    tv = importlib.import_module("diffbir.utils.tilevae.tilevae")
    cldm, swin, diff, W = build_reference(R, "tiny", configs.get("DIFFUSION_V21"))
    x = torch.tensor(cases.make_lq(31, 1, 608, 712)).float().div(255).permute(0, 3, 1, 2).contiguous()
    g = {}
    with cases.quiet():
        g["enc_tiled_256"] = cldm.vae_encode(x * 2 - 1, sample=False, tiled=True, tile_size=256).numpy()
        z = cases.NoiseStream(9)((1, 4, 76, 89))
        g["dec_tiled_32"] = cldm.vae_decode(z, tiled=True, tile_size=32).numpy()
        x2 = torch.tensor(cases.make_lq(32, 2, 384, 320)).float().div(255).permute(0, 3, 1, 2).contiguous()
        g["enc_tiled_b2_128"] = cldm.vae_encode(x2 * 2 - 1, sample=False, tiled=True, tile_size=128).numpy()
        # whole pipeline with --vae_encoder_tiled --vae_decoder_tiled (tile sizes 256 / 256), 3 spaced steps
        pipe = R.SwinIRPipeline(swin, cldm, diff, None, "cpu")
        torch.manual_seed(5)
        g["pipe_vae_tiled"] = pipe.run(cases.make_lq(9, 1, 600, 712), 3, 1.0, False, 512, 256, True, 256, True, 256, False, 512,
                                       256, "", cases.NEG_PROMPT, 4.0, "noise", "spaced", 0, False, 0, 0, 300, 1, 1, 1)
    np.savez_compressed(os.path.join(OUT, "tiny_tiled_vae.npz"), **g)
    geo = {}
    for (h, w, ts, dec) in ((608, 712, 256, False), (76, 89, 32, True), (2048, 2048, 256, False), (256, 256, 32, True),
                            (384, 320, 128, False), (600, 200, 256, False), (512, 512, 64, True), (97, 131, 32, True)):
        hook = tv.VAEHook(None, ts, dec, False, False, False)
        with cases.quiet():
            ins, outs = hook.split_tiles(h, w)
        geo[f"{h}x{w}_{ts}_{'dec' if dec else 'enc'}"] = dict(ins=[list(map(int, b)) for b in ins],
                                                                outs=[list(map(int, b)) for b in outs])
    with open(os.path.join(OUT, "tiled_vae_geometry.json"), "w") as f:
        json.dump(geo, f)
    print("tiled vae done", {k: v.shape for k, v in g.items()}, len(geo))


TOKENIZER_PROMPTS = [
    "", "low quality, blurry, low-resolution, noisy, unsharp, weird textures", "a photo of a cat",
    "Cinematic, High Contrast, highly detailed, taken using a Canon EOS R camera, hyper detailed photo - realistic "
    "maximum detail, 32k, Color Grading, ultra HD, extreme meticulous detailing, skin pore detailing, hyper sharpness, "
    "perfect without deformations.",
    "painting, oil painting, illustration, drawing, art, sketch, oil painting, cartoon, CG Style, 3D render, unreal "
    "engine, blurring, dirty, messy, worst quality, low quality, frames, watermark, signature, jpeg artifacts, "
    "deformed, lowres, over-smooth.",
    "A close-up portrait of an elderly man's face, wrinkles & freckles; 85mm f/1.4", "it's the dog's ball, they're here",
    "we've   got    multiple\tspaces\nand newlines", "UPPERCASE and MiXeD Case", "numbers 1234567890 and 3.14159",
    "hyphen-ated under_score slash/back\\slash", "emoji \U0001F600 and accents caf\u00e9 na\u00efve \u00fcber",
    "&amp; html &lt;entities&gt; &quot;quoted&quot;", "<start_of_text> literal special <end_of_text>",
    "\u4e2d\u6587 \u65e5\u672c\u8a9e \ud55c\uad6d\uc5b4", "supercalifragilisticexpialidocious antidisestablishmentarianism",
    "a " * 100, "!!! ??? ... ,,, ;;; :::", "I'd've'll'm're's't", "x" * 300, "street at night, neon signs, rain, 4k",
    "the quick brown fox jumps over the lazy dog", "\u00bd \u00be \u2122 \u00a9 \u20ac \u00a3 \u00a5",
]


def gen_tokenizer(R):
    """Token ids of the reference tokenizer (open_clip/tokenizer.py) for a set of prompts covering contractions,
    whitespace clean-up, html entities, non-ASCII, special tokens and truncation at 77."""
    import json
    from diffbir.model.open_clip import tokenize
    ids = tokenize(TOKENIZER_PROMPTS).tolist()
    with open(os.path.join(OUT, "tokenizer_cases.json"), "w") as f:
        json.dump(dict(prompts=TOKENIZER_PROMPTS, ids=ids), f)
    print("tokenizer cases", len(ids))


# DDIM + EDM / k-diffusion samplers on the tiny config: name -> (sampler, steps, extra run args)
SAMPLER_CASES = {
    "ddim8": ("ddim", 8, {}),
    "ddim5_rescale": ("ddim", 5, dict(rescale_cfg=True, cfg=3.0)),
    "edm_euler": ("edm_euler", 6, {}),
    "edm_euler_churn": ("edm_euler", 6, dict(s_churn=4.0, s_tmin=0.05, s_tmax=50.0, s_noise=1.003)),
    "edm_euler_a": ("edm_euler_a", 6, {}),
    "edm_heun": ("edm_heun", 5, {}),
    "edm_dpm_2": ("edm_dpm_2", 5, {}),
    "edm_dpm_2_a": ("edm_dpm_2_a", 5, {}),
    "edm_lms": ("edm_lms", 7, dict(order=3)),
    "edm_dpm++_2s_a": ("edm_dpm++_2s_a", 5, {}),
    "edm_dpm++_2m": ("edm_dpm++_2m", 8, {}),
    # SDE solvers: torchsde is not installed here, so k_diffusion.BrownianTreeNoiseSampler is replaced ON BOTH SIDES by
    # a stand-in that returns a fresh N(0, 1) draw per call (solver arithmetic under test, not the Brownian tree)
    "edm_dpm++_sde": ("edm_dpm++_sde", 5, {}),
    "edm_dpm++_2m_sde": ("edm_dpm++_2m_sde", 8, {}),
    "edm_dpm++_3m_sde": ("edm_dpm++_3m_sde", 10, {}),
    "edm_dpm++_3m_sde_eps": ("edm_dpm++_3m_sde", 6, dict(version="v2")),
}


class _IidNoiseSampler:
    """Stand-in for k_diffusion.BrownianTreeNoiseSampler in the golden generator (see SAMPLER_CASES)."""

    def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda x: x):
        self.x = x

    def __call__(self, sigma, sigma_next):
        return torch.randn_like(self.x)


@torch.no_grad()
def gen_samplers(R):
    import importlib
    from diffbir_amd import configs
    kd = importlib.import_module("diffbir.sampler.k_diffusion")
    kd.BrownianTreeNoiseSampler = _IidNoiseSampler
    g = {}
    built = {}
    for name, (sampler, steps, kw) in SAMPLER_CASES.items():
        kw = dict(kw)
        ver = kw.pop("version", "v21")
        if ver not in built:
            built[ver] = build_reference(R, "tiny", configs.get("DIFFUSION_V21" if ver == "v21" else "DIFFUSION_V2"))
        cldm, swin, diff, W = built[ver]
        pipe = R.SwinIRPipeline(swin, cldm, diff, None, "cpu")
        torch.manual_seed(17)
        a = dict(cfg=4.0, rescale_cfg=False, s_churn=0, s_tmin=0, s_tmax=300, s_noise=1, eta=1, order=1)
        a.update(kw)
        with cases.quiet():
            g[name] = pipe.run(cases.make_lq(3, 1, 512, 512), steps, 1.0, False, 512, 256, False, 256, False, 256, False,
                               512, 256, "", cases.NEG_PROMPT, a["cfg"], "noise", sampler, 0, a["rescale_cfg"],
                               a["s_churn"], a["s_tmin"], a["s_tmax"], a["s_noise"], a["eta"], a["order"])
        print(name, g[name].shape, flush=True)
    np.savez_compressed(os.path.join(OUT, "tiny_samplers.npz"), **g)


# SDE solvers on the reference's OWN BrownianTreeNoiseSampler / BatchedBrownianTree (k_diffusion.py:70-119), running on the
# restated torchsde tree of oracle/refshim/torchsde (parity of that tree with a real torchsde install is unpinned — its header
# says so): what these fixtures pin is everything the reference does around the tree — the seed drawn from the global CPU
# generator, sigma_min / sigma_max, the query times of each solver, sign and 1/sqrt|dt| normalisation — and that the engine's
# independently written tree (diffbir_amd/sampler/brownian.py) realises the same noise.  name -> (sampler, steps, lq spec)
SAMPLER_TREE_CASES = {
    "edm_dpm++_sde": ("edm_dpm++_sde", 5, (3, 1, 512, 512)),
    "edm_dpm++_2m_sde": ("edm_dpm++_2m_sde", 8, (3, 1, 512, 512)),
    "edm_dpm++_3m_sde": ("edm_dpm++_3m_sde", 10, (3, 1, 512, 512)),
    "edm_dpm++_3m_sde_b2": ("edm_dpm++_3m_sde", 6, (5, 2, 512, 512)),
}


@torch.no_grad()
def gen_samplers_tree(R):
    import importlib
    from diffbir_amd import configs
    kd = importlib.import_module("diffbir.sampler.k_diffusion")
    assert kd.BrownianTreeNoiseSampler.__module__ == kd.__name__, "the reference's own noise sampler must be in place"
    assert kd.torchsde.__version__.endswith("restated")
    cldm, swin, diff, W = build_reference(R, "tiny", configs.get("DIFFUSION_V21"))
    g = {}
    for name, (sampler, steps, lq) in SAMPLER_TREE_CASES.items():
        pipe = R.SwinIRPipeline(swin, cldm, diff, None, "cpu")
        torch.manual_seed(17)
        with cases.quiet():
            g[name] = pipe.run(cases.make_lq(*lq), steps, 1.0, False, 512, 256, False, 256, False, 256, False, 512, 256, "",
                               cases.NEG_PROMPT, 4.0, "noise", sampler, 0, False, 0, 0, 300, 1, 1, 1)
        print(name, g[name].shape, flush=True)
    np.savez_compressed(os.path.join(OUT, "tiny_samplers_tree.npz"), **g)


TINY_PIPE_CASES = {   # tiny end-to-end cases of tests/golden/tiny_pipeline.npz (v2.1 schedule)
    "spaced6_v21": ((3, 1, 512, 512), 6, "spaced", 231, {}),
    "dpm10_v21": ((3, 1, 512, 512), 10, "dpm++_m2", 231, {}),
    "spaced4_b2_v21": ((5, 2, 512, 512), 4, "spaced", 99, {}),
    "spaced3_pad_v21": ((9, 1, 600, 712), 3, "spaced", 5, {}),
    "spaced3_tiled_v21": ((9, 1, 600, 712), 3, "spaced", 5, dict(tiled=True)),
    "dpm10_tiled_v21": ((9, 1, 600, 712), 10, "dpm++_m2", 5, dict(tiled=True)),
}


@torch.no_grad()
def gen_ref_lowp(R):
    """PSNR of the REFERENCE's own reduced-precision paths (cast_dtype + torch.autocast on CPU, what `--precision
    fp16 / bf16` selects: loop.py:86-96,180) against its fp32 output, per tiny case: the yardstick for the engine's bf16
    tolerance (bf16 carries 8 mantissa bits against fp16's 11: ~18 dB less, for the reference as for the engine)."""
    import json
    from diffbir_amd import configs
    g = np.load(os.path.join(OUT, "tiny_pipeline.npz"))
    res = {}
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        cldm, swin, diff, W = build_reference(R, "tiny", configs.get("DIFFUSION_V21"))
        cldm.cast_dtype(dt)
        for name, (lqspec, steps, sampler, seed, kw) in TINY_PIPE_CASES.items():
            with torch.autocast("cpu", dt):
                out = run_pipeline(R, cldm, swin, diff, cases.make_lq(*lqspec), steps, sampler, seed, **kw)
            res[f"{name}_{tag}"] = cases.psnr_u8(out, g[name])
            print(name, tag, f"{res[f'{name}_{tag}']:.2f} dB", flush=True)
    # second-order EDM solvers start from sigma_0 = 1e4 (alphas_cumprod[0] := 1e-8) and amplify the 16-bit error of their
    # second network evaluation by dt / sigma_next ~ 500x in the first step; eps-parameterisation at sigma = 1e4 cancels
    # x - sigma*eps completely in fp16.  The reference's own fp16 run documents what any fp16 implementation can reach.
    import importlib
    kd = importlib.import_module("diffbir.sampler.k_diffusion")
    kd.BrownianTreeNoiseSampler = _IidNoiseSampler
    gs = np.load(os.path.join(OUT, "tiny_samplers.npz"))
    for name in ("edm_heun", "edm_dpm_2", "edm_dpm_2_a", "edm_dpm++_3m_sde_eps", "edm_euler", "edm_dpm++_2m"):
        sampler, steps, kw = SAMPLER_CASES[name]
        kw = dict(kw)
        ver = kw.pop("version", "v21")
        cldm, swin, diff, W = build_reference(R, "tiny", configs.get("DIFFUSION_V21" if ver == "v21" else "DIFFUSION_V2"))
        cldm.cast_dtype(torch.float16)
        pipe = R.SwinIRPipeline(swin, cldm, diff, None, "cpu")
        torch.manual_seed(17)
        a = dict(cfg=4.0, rescale_cfg=False, s_churn=0, s_tmin=0, s_tmax=300, s_noise=1, eta=1, order=1)
        a.update(kw)
        with cases.quiet(), torch.autocast("cpu", torch.float16):
            out = pipe.run(cases.make_lq(3, 1, 512, 512), steps, 1.0, False, 512, 256, False, 256, False, 256, False, 512,
                           256, "", cases.NEG_PROMPT, a["cfg"], "noise", sampler, 0, a["rescale_cfg"], a["s_churn"],
                           a["s_tmin"], a["s_tmax"], a["s_noise"], a["eta"], a["order"])
        res[f"sampler_{name}_fp16"] = cases.psnr_u8(out, gs[name])
        print("sampler", name, "fp16", f"{res[f'sampler_{name}_fp16']:.2f} dB", flush=True)
    with open(os.path.join(OUT, "reference_lowp_psnr.json"), "w") as f:
        json.dump(res, f, indent=1)


def gen_host_tables(R):
    """Host-side tables of the path, straight from the reference's own functions: timestep spacing, the spaced
    sampler's registered buffers, the DPM-Solver discrete VP schedule, tile windows and blend weights."""
    import importlib
    import json
    from diffbir_amd import configs
    sp = importlib.import_module("diffbir.sampler.spaced_sampler")
    dp = importlib.import_module("diffbir.sampler.dpm_solver_pytorch")
    g = {"space_timesteps": {}, "spaced_tables": {}, "dpm": {}, "sliding_windows": {}, "gaussian_weights": {}}
    for n in (1, 2, 3, 5, 7, 10, 20, 25, 50, 100, 250, 999, 1000):
        g["space_timesteps"][str(n)] = sorted(int(x) for x in sp.space_timesteps(1000, str(n)))
    g["space_timesteps"]["ddim50"] = sorted(int(x) for x in sp.space_timesteps(1000, "ddim50"))
    for name in ("DIFFUSION_V2", "DIFFUSION_V21"):
        diff = R.Diffusion(**configs.get(name))
        for steps in (5, 50):
            smp = R.SpacedSampler(diff.betas, diff.parameterization, rescale_cfg=False)
            with np.errstate(divide="ignore", invalid="ignore"):
                smp.make_schedule(steps)
            tb = {k: getattr(smp, k).double().numpy() for k in (
                "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_mean_coef1", "posterior_mean_coef2")}
            g["spaced_tables"][f"{name}_{steps}"] = {
                "timesteps": [int(t) for t in smp.timesteps],
                **{k: [None if not np.isfinite(x) else float(x) for x in v] for k, v in tb.items()}}
        ns = dp.NoiseScheduleVP("discrete", betas=torch.tensor(diff.betas, dtype=torch.float32))
        solver = dp.DPM_Solver(lambda x, t: x, ns, algorithm_type="dpmsolver++")
        ts = solver.get_time_steps("time_uniform", ns.T, 1.0 / ns.total_N, 20, "cpu")
        g["dpm"][name] = dict(total_N=int(ns.total_N), T=float(ns.T), t=[float(x) for x in ts],
                              alpha=[float(x) for x in ns.marginal_alpha(ts)],
                              std=[float(x) for x in ns.marginal_std(ts)],
                              lam=[float(x) for x in ns.marginal_lambda(ts)],
                              model_t=[float((x - 1.0 / ns.total_N) * 1000.0) for x in ts])
    for (h, w, size, stride) in ((64, 64, 64, 32), (75, 89, 64, 32), (256, 256, 64, 32), (100, 64, 64, 48),
                                 (600, 712, 512, 256)):
        g["sliding_windows"][f"{h}x{w}_{size}_{stride}"] = [list(map(int, win))
                                                             for win in R.common.sliding_windows(h, w, size, stride)]
    for (tw, th) in ((64, 64), (512, 512), (32, 48)):
        gw = R.common.gaussian_weights(tw, th).astype(np.float64)
        if gw.size <= 4096:
            g["gaussian_weights"][f"{tw}x{th}"] = dict(full=gw.tolist())
        else:  # large tiles: a few rows / columns + the total (the weights are separable)
            g["gaussian_weights"][f"{tw}x{th}"] = dict(
                shape=list(gw.shape), total=float(gw.sum()),
                rows={str(r): gw[r].tolist() for r in (0, th // 2 - 1, th // 2, th - 1)},
                cols={str(c): gw[:, c].tolist() for c in (0, tw // 2 - 1, tw // 2, tw - 1)})
    with open(os.path.join(OUT, "host_tables.json"), "w") as f:
        json.dump(g, f)
    print("host tables done", {k: len(v) for k, v in g.items()})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    what = sys.argv[1]
    R = load_reference()
    from diffbir_amd import configs
    if what == "tiny":
        gen_modules(R, "tiny", "tiny", 128, configs.get("DIFFUSION_V21"))
        gen_tiny_pipelines(R)
    elif what == "host_tables":
        gen_host_tables(R)
    elif what == "tiny_options":
        gen_tiny_options(R)
    elif what == "full_modules":
        gen_modules(R, "full", "full", 256, configs.get("DIFFUSION_V21"))
    elif what == "full_pipeline":
        gen_full_pipeline(R)
    elif what == "tiled_vae":
        gen_tiled_vae(R)
    elif what == "samplers":
        gen_samplers(R)
    elif what == "samplers_tree":
        gen_samplers_tree(R)
    elif what == "ref_lowp":
        gen_ref_lowp(R)
    elif what == "tokenizer":
        gen_tokenizer(R)
    elif what == "cleaners":
        gen_cleaners(R)
    elif what == "full_configs":
        gen_full_configs(R, only=sys.argv[2:] or None)
    elif what == "full_bf16":
        gen_full_bf16(R)
    elif what == "full_extra":
        gen_full_extra(R)
