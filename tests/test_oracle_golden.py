"""Pins the oracle (CPU restatement) to golden vectors produced by the unmodified reference
(oracle/make_golden.py).  Runs everywhere (no reference checkout, no GPU needed)."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, nets, sampling
from oracle.pipeline import OraclePipeline
from diffbir_amd import configs


@pytest.fixture(scope="module")
def tiny():
    cldm_cfg, swin_cfg = cases.get_cfgs("tiny")
    return cldm_cfg, swin_cfg, cases.synth_weights(cldm_cfg, swin_cfg, 0)


@pytest.fixture(scope="module")
def gm(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "tiny_modules.npz")))


def _close(a, b, tol):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    err = (a - b).abs().max().item()
    scale = b.abs().max().item()
    assert err <= tol * max(scale, 1.0), f"max abs err {err} (scale {scale})"


@torch.no_grad()
def test_modules_match_reference_golden(tiny, gm):
    cldm_cfg, swin_cfg, W = tiny
    rs = cases.NoiseStream(7)
    x = torch.tensor(cases.make_lq(11, 2, 128, 128)).float().div(255).permute(0, 3, 1, 2).contiguous()
    _close(nets.swinir_forward(W["swinir"], swin_cfg, x), gm["swinir_out"], 2e-5)
    _close(nets.vae_encode_mode(W["vae"], cldm_cfg["vae_cfg"], x * 2 - 1, 0.18215), gm["vae_mode"], 2e-5)
    z = rs((2, 4, 16, 16))
    _close(nets.vae_decode(W["vae"], cldm_cfg["vae_cfg"], z, 0.18215), gm["vae_dec"], 2e-5)
    toks = torch.tensor(gm["tokens"])
    c_txt = nets.clip_text_encode(W["clip"], cldm_cfg["clip_cfg"], toks)
    _close(c_txt, gm["c_txt"], 2e-5)
    xn = rs((2, 4, 16, 16))
    c_img = rs((2, 4, 16, 16)) * 0.5
    sc = [0.9] * 13
    e = nets.cldm_forward(W, cldm_cfg, xn, torch.tensor([999, 381]), c_txt, c_img, sc)
    _close(e, gm["eps_int_t"], 5e-5)
    e = nets.cldm_forward(W, cldm_cfg, xn, torch.tensor([949.0365, 49.95]), c_txt, c_img, sc)
    _close(e, gm["eps_float_t"], 5e-5)
    ctrl = nets.controlnet_forward(W["controlnet"], cldm_cfg["controlnet_cfg"], xn, c_img,
                                   torch.tensor([999, 381]), c_txt)
    assert len(ctrl) == 13
    _close(ctrl[0], gm["control_0"], 5e-5)
    _close(ctrl[12], gm["control_12"], 5e-5)


def _pipe(tiny, gm, dcfg):
    cldm_cfg, swin_cfg, W = tiny
    table = {"": torch.tensor(gm["tokens"][0]), cases.NEG_PROMPT: torch.tensor(gm["tokens"][1])}
    return OraclePipeline(W, cldm_cfg, swin_cfg, configs.get(dcfg),
                          tokenize=lambda txts: torch.stack([table[t] for t in txts]))


CASES = [
    ("spaced6_v21", "DIFFUSION_V21", dict(lq=(3, 1, 512, 512), steps=6, sampler_type="spaced", seed=231)),
    ("spaced6_v2", "DIFFUSION_V2", dict(lq=(3, 1, 512, 512), steps=6, sampler_type="spaced", seed=231)),
    ("dpm10_v21", "DIFFUSION_V21", dict(lq=(3, 1, 512, 512), steps=10, sampler_type="dpm++_m2", seed=231)),
    ("dpm10_v2", "DIFFUSION_V2", dict(lq=(3, 1, 512, 512), steps=10, sampler_type="dpm++_m2", seed=231)),
    ("spaced4_b2_v21", "DIFFUSION_V21", dict(lq=(5, 2, 512, 512), steps=4, sampler_type="spaced", seed=99)),
    ("spaced3_pad_v21", "DIFFUSION_V21", dict(lq=(9, 1, 600, 712), steps=3, sampler_type="spaced", seed=5)),
    ("spaced3_tiled_v21", "DIFFUSION_V21", dict(lq=(9, 1, 600, 712), steps=3, sampler_type="spaced", seed=5,
                                                 cldm_tiled=True)),
    ("dpm10_tiled_v21", "DIFFUSION_V21", dict(lq=(9, 1, 600, 712), steps=10, sampler_type="dpm++_m2", seed=5,
                                               cldm_tiled=True)),
]


@pytest.mark.parametrize("name,dcfg,kw", CASES, ids=[c[0] for c in CASES])
def test_pipeline_matches_reference_golden(tiny, gm, golden_dir, name, dcfg, kw):
    gp = np.load(os.path.join(golden_dir, "tiny_pipeline.npz"))
    kw = dict(kw)
    lq = cases.make_lq(*kw.pop("lq"))
    seed = kw.pop("seed")
    out = _pipe(tiny, gm, dcfg).run(lq, neg_prompt=cases.NEG_PROMPT, cfg_scale=4.0,
                                     randn=cases.NoiseStream(seed), **kw)
    ref = gp[name]
    assert out.shape == ref.shape
    diff = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    # fp32 CPU on both sides: identical up to rounding at the u8 quantisation boundary
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())
    assert cases.psnr_u8(out, ref) > 70.0


OPTION_KW = {
    "cond_start": dict(lq=(3, 1, 512, 512), steps=4, sampler_type="spaced", seed=7, start_point_type="cond"),
    "noise_aug": dict(lq=(3, 1, 512, 512), steps=4, sampler_type="spaced", seed=7, noise_aug=120),
    "rescale_cfg": dict(lq=(3, 1, 512, 512), steps=4, sampler_type="spaced", seed=7, rescale_cfg=True, cfg_scale=3.0),
    "cfg1": dict(lq=(3, 1, 512, 512), steps=4, sampler_type="spaced", seed=7, cfg_scale=1.0),
    "strength": dict(lq=(3, 1, 512, 512), steps=4, sampler_type="dpm++_m2", seed=7, strength=0.6),
    "cleaner_tiled": dict(lq=(9, 1, 600, 712), steps=3, sampler_type="spaced", seed=5, cleaner_tiled=True),
    "small_upsized": dict(lq=(13, 1, 300, 256), steps=3, sampler_type="spaced", seed=5),
}


@pytest.mark.parametrize("name", sorted(OPTION_KW))
def test_pipeline_options_match_reference_golden(tiny, gm, golden_dir, name):
    """the oracle's option paths (pipeline.py:146-174, 371-397; sampler.py:31-38) pinned to the reference itself."""
    ref = np.load(os.path.join(golden_dir, "tiny_options.npz"))[name]
    kw = dict(OPTION_KW[name])
    lq = cases.make_lq(*kw.pop("lq"))
    seed = kw.pop("seed")
    kw.setdefault("cfg_scale", 4.0)
    out = _pipe(tiny, gm, "DIFFUSION_V21").run(lq, neg_prompt=cases.NEG_PROMPT, randn=cases.NoiseStream(seed), **kw)
    assert out.shape == ref.shape
    diff = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())


@torch.no_grad()
@pytest.mark.parametrize("name", sorted(cases.CLEANERS))
def test_cleaners_match_reference_golden(golden_dir, name):
    """oracle RRDBNet / SCUNet restatements == outputs of the reference's own modules (tests/golden/cleaners.npz)."""
    g = np.load(os.path.join(golden_dir, "cleaners.npz"))
    cfg, W, x = cases.cleaner_case(name)
    fn = nets.rrdbnet_forward if name.startswith("bsrnet") else nets.scunet_forward
    _close(fn(W, cfg, x), g[name], 2e-5)


def test_golden_manifest(golden_dir):
    """Every fixture under tests/golden/ is listed in MANIFEST.json with its sha256 (tools/golden_manifest.py): a fixture
    cannot change silently, and each entry records the generator sources (oracle/make_golden.py, oracle/cases.py, ...) it
    was produced with."""
    import hashlib
    import json
    with open(os.path.join(golden_dir, "MANIFEST.json")) as f:
        man = json.load(f)
    names = sorted(n for n in os.listdir(golden_dir) if n != "MANIFEST.json")
    assert names == sorted(man["files"]), set(names) ^ set(man["files"])
    for n in names:
        h = hashlib.sha256(open(os.path.join(golden_dir, n), "rb").read()).hexdigest()
        assert h == man["files"][n]["sha256"], f"{n} changed without `python tools/golden_manifest.py`"
        g = man["files"][n]["generator"]
        assert "note" in g or set(g) >= {"oracle/make_golden.py", "oracle/cases.py"}
