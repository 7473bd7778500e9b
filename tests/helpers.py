"""Shared test helpers: build the engine-side models from seeded synthetic weights."""
import numpy as np
import torch

from oracle import cases
from diffbir_amd import configs
from diffbir_amd.model import ControlLDM, Diffusion, SwinIR
from diffbir_amd.pipeline import SwinIRPipeline


def build_engine(cfg_name: str, diffusion_cfg: str, device, dtype, W=None, raw_dtype=False):
    """-> (pipeline, cldm, swinir). raw_dtype=True forces `_dtype` without the 16-bit check (CPU emulation in f32)."""
    cldm_cfg, swin_cfg = cases.get_cfgs(cfg_name)
    if W is None:
        W = cases.synth_weights(cldm_cfg, swin_cfg, 0)
    cldm = ControlLDM(**cldm_cfg)
    cldm.unet.load_state_dict(W["unet"], strict=True)
    cldm.controlnet.load_state_dict(W["controlnet"], strict=True)
    cldm.vae.load_state_dict(W["vae"], strict=True)
    cldm.clip.load_state_dict(W["clip"], strict=True)
    swin = SwinIR(**swin_cfg)
    swin.load_state_dict(W["swinir"], strict=True)
    mods = [cldm.unet, cldm.controlnet, cldm.vae, cldm.clip, swin]
    for m in mods:
        m.to(device)
    for m in mods:
        if raw_dtype:
            m._dtype, m._packed = dtype, False
        else:
            m.set_dtype(dtype)
    diff = Diffusion(**configs.get(diffusion_cfg))
    pipe = SwinIRPipeline(swin, cldm, diff, None, str(device))
    return pipe, cldm, swin


def run_pipe(pipe, lq, steps, sampler, seed, cfg=4.0, tiled=False, tile=512, stride=256, cleaner_tiled=False,
             strength=1.0, start="noise", noise_aug=0, rescale_cfg=False, vae_tiled=False):
    pipe.randn = cases.NoiseStream(seed)
    return pipe.run(lq, steps, strength, cleaner_tiled, 512, 256, vae_tiled, 256, vae_tiled, 256, tiled, tile, stride,
                    "", cases.NEG_PROMPT, cfg, start, sampler, noise_aug, rescale_cfg, 0, 0, 300, 1, 1, 1)


# option paths of Pipeline.run (goldens: tests/golden/tiny_options.npz, oracle/make_golden.py OPTION_CASES)
OPTION_CASES = {
    "cond_start": ((3, 1, 512, 512), dict(steps=4, sampler="spaced", seed=7, start="cond")),
    "noise_aug": ((3, 1, 512, 512), dict(steps=4, sampler="spaced", seed=7, noise_aug=120)),
    "rescale_cfg": ((3, 1, 512, 512), dict(steps=4, sampler="spaced", seed=7, rescale_cfg=True, cfg=3.0)),
    "cfg1": ((3, 1, 512, 512), dict(steps=4, sampler="spaced", seed=7, cfg=1.0)),
    "strength": ((3, 1, 512, 512), dict(steps=4, sampler="dpm++_m2", seed=7, strength=0.6)),
    "cleaner_tiled": ((9, 1, 600, 712), dict(steps=3, sampler="spaced", seed=5, cleaner_tiled=True)),
    "small_upsized": ((13, 1, 300, 256), dict(steps=3, sampler="spaced", seed=5)),
}


def run_option_case(pipe, name):
    lqspec, kw = OPTION_CASES[name]
    kw = dict(kw)
    return run_pipe(pipe, cases.make_lq(*lqspec), kw.pop("steps"), kw.pop("sampler"), kw.pop("seed"), **kw)


# DDIM / EDM sampler cases (goldens: tests/golden/tiny_samplers.npz, oracle/make_golden.py SAMPLER_CASES)
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
    "edm_dpm++_sde": ("edm_dpm++_sde", 5, {}),
    "edm_dpm++_2m_sde": ("edm_dpm++_2m_sde", 8, {}),
    "edm_dpm++_3m_sde": ("edm_dpm++_3m_sde", 10, {}),
    "edm_dpm++_3m_sde_eps": ("edm_dpm++_3m_sde", 6, dict(version="v2")),
}


def run_sampler_case(pipe, name):
    """Same call as oracle/make_golden.py gen_samplers; the SDE solvers get the i.i.d. stand-in for the Brownian tree
    that the golden generator installed in the reference (torchsde is absent on both sides)."""
    sampler, steps, kw = SAMPLER_CASES[name]
    a = dict(cfg=4.0, rescale_cfg=False, s_churn=0, s_tmin=0, s_tmax=300, s_noise=1, eta=1, order=1)
    a.update({k: v for k, v in kw.items() if k != "version"})
    pipe.randn = cases.NoiseStream(17)
    pipe.brownian = lambda x, randn, *_range: (lambda sigma, sigma_next: randn(tuple(x.shape)))
    try:
        return pipe.run(cases.make_lq(3, 1, 512, 512), steps, 1.0, False, 512, 256, False, 256, False, 256, False, 512, 256,
                        "", cases.NEG_PROMPT, a["cfg"], "noise", sampler, 0, a["rescale_cfg"], a["s_churn"], a["s_tmin"],
                        a["s_tmax"], a["s_noise"], a["eta"], a["order"])
    finally:
        pipe.brownian = None


# the SDE solvers on the restated torchsde tree (goldens: tests/golden/tiny_samplers_tree.npz, oracle/make_golden.py)
SAMPLER_TREE_CASES = {
    "edm_dpm++_sde": ("edm_dpm++_sde", 5, (3, 1, 512, 512)),
    "edm_dpm++_2m_sde": ("edm_dpm++_2m_sde", 8, (3, 1, 512, 512)),
    "edm_dpm++_3m_sde": ("edm_dpm++_3m_sde", 10, (3, 1, 512, 512)),
    "edm_dpm++_3m_sde_b2": ("edm_dpm++_3m_sde", 6, (5, 2, 512, 512)),
}


def run_sampler_tree_case(pipe, name):
    """Same call as oracle/make_golden.py gen_samplers_tree.  The engine's own Brownian tree (sampler/brownian.py), with the two
    things the reference takes from torch's global CPU RNG state tied to the test's noise stream instead: the tree seed
    (k_diffusion.py:78-79, drawn after x_T) and the per-node generators (the golden ran on device = cpu)."""
    from diffbir_amd.sampler.brownian import BrownianTreeNoise
    sampler, steps, lq = SAMPLER_TREE_CASES[name]
    stream = cases.NoiseStream(17)
    pipe.randn = stream
    pipe.brownian = lambda x, randn, smin, smax: BrownianTreeNoise(
        x, smin, smax, seed=int(torch.randint(0, 2 ** 63 - 1, [], generator=stream.g).item()), noise_device="cpu")
    try:
        return pipe.run(cases.make_lq(*lq), steps, 1.0, False, 512, 256, False, 256, False, 256, False, 512, 256, "",
                        cases.NEG_PROMPT, 4.0, "noise", sampler, 0, False, 0, 0, 300, 1, 1, 1)
    finally:
        pipe.brownian = None


def rel_err(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item(), (a - b).abs().max().item()


# ---- BSRNet / SCUNet cleaners (goldens: tests/golden/cleaners.npz, oracle/make_golden.py gen_cleaners) ----------------
def build_cleaner(name: str, device, dtype, raw_dtype=False):
    """-> (engine module with the seeded weights of oracle.cases.cleaner_case(name), input f32 NCHW)."""
    from diffbir_amd.model import RRDBNet, SCUNet
    cfg, W, x = cases.cleaner_case(name)
    m = (RRDBNet if name.startswith("bsrnet") else SCUNet)(**cfg)
    m.load_state_dict(W, strict=True)
    m.to(device)
    if raw_dtype:
        m._dtype, m._packed = dtype, False
    else:
        m.set_dtype(dtype)
    return m, x.to(device)


def run_cleaner_pipeline(name: str, cldm, diff, device, dtype, raw_dtype=False):
    """The BSRNetPipeline / SCUNetPipeline run of oracle.cases.CLEANER_PIPELINES[name] on the engine."""
    from diffbir_amd.pipeline import BSRNetPipeline, SCUNetPipeline
    cleaner, lqspec, kw = cases.CLEANER_PIPELINES[name]
    m, _ = build_cleaner(cleaner, device, dtype, raw_dtype)
    if cleaner.startswith("bsrnet"):
        pipe = BSRNetPipeline(m, cldm, diff, None, str(device), kw["upscale"])
    else:
        pipe = SCUNetPipeline(m, cldm, diff, None, str(device))
    pipe.randn = cases.NoiseStream(kw["seed"])
    return pipe.run(cases.make_lq(*lqspec), kw["steps"], 1.0, kw.get("cleaner_tiled", False), kw.get("cleaner_tile", 512),
                    kw.get("cleaner_stride", 256), False, 256, False, 256, False, 512, 256, "", cases.NEG_PROMPT, 4.0,
                    "noise", "spaced", 0, False, 0, 0, 300, 1, 1, 1)
