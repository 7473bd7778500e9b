"""Inference loops with the reference's class surface (reference diffbir/inference/loop.py:32-236): the same hook
methods (`load_cleaner`, `load_cldm`, `load_cond_fn`, `load_pipeline`, `load_captioner`, `setup`, `load_lq`,
`after_load_lq`, `run`, `save`), driven by the same `argparse.Namespace` as the reference CLI (inference.py).

Engine differences, all on the control plane: configs come from `diffbir_amd.configs` (or the YAML files under
`configs/inference/` when present; PyYAML instead of OmegaConf), there is no autocast context (precision is a property
of the engine: `cldm.cast_dtype`), captioners other than `none` and restoration guidance are outside this engine's scope
 and are refused with the reference's own error style.
"""
import os
from argparse import Namespace
from typing import Generator, List

import numpy as np
import torch

from .. import configs
from ..model import ControlLDM, Diffusion
from ..pipeline import Pipeline
from ..utils.common import instantiate_from_config, load_model_from_url
from .pretrained_models import MODELS

_CFG_DIR = os.path.join("configs", "inference")


def load_config(name: str) -> dict:
    """`configs/inference/<name>.yaml` relative to the working directory if it exists (what the reference reads), else
    the built-in copy of the same values (diffbir_amd/configs.py)."""
    path = os.path.join(_CFG_DIR, name + ".yaml")
    if os.path.exists(path):
        import yaml
        with open(path) as f:
            return yaml.safe_load(f)
    return configs.yaml_config(name)


class EmptyCaptioner:
    """reference utils/caption.py:44-52."""

    def __init__(self, device=None):
        self.device = device

    def __call__(self, image) -> str:
        return ""


class InferenceLoop:
    def __init__(self, args: Namespace) -> None:
        self.args = args
        self.loop_ctx = {}
        self.pipeline: Pipeline = None
        self.load_cleaner()
        self.load_cldm()
        self.load_cond_fn()
        self.load_pipeline()
        self.load_captioner()

    def load_cleaner(self) -> None:  # pragma: no cover - abstract
        raise NotImplementedError

    def load_cldm(self) -> None:
        """reference loop.py:48-96."""
        self.cldm: ControlLDM = instantiate_from_config(load_config("cldm"))
        sd_weight = load_model_from_url(MODELS["sd_v2.1_zsnr" if self.args.version == "v2.1" else "sd_v2.1"])
        unused, missing = self.cldm.load_pretrained_sd(sd_weight)
        print(f"load pretrained stable diffusion, unused weights: {unused}, missing weights: {missing}")
        if self.args.version == "v1":
            if self.args.task == "face":
                control_weight = load_model_from_url(MODELS["v1_face"])
            elif self.args.task in ("sr", "denoise"):
                control_weight = load_model_from_url(MODELS["v1_general"])
            else:
                raise ValueError(f"DiffBIR v1 doesn't support task: {self.args.task}, "
                                 f"please use v2 or v2.1 by passsing '--version'")
        elif self.args.version == "v2":
            control_weight = load_model_from_url(MODELS["v2"])
        else:
            control_weight = load_model_from_url(MODELS["v2.1"])
        self.cldm.load_controlnet_from_ckpt(control_weight)
        print("load controlnet weight")
        self.cldm.eval().to(self.args.device)
        cast_type = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[self.args.precision]
        self.cldm.cast_dtype(cast_type)
        name = "diffusion" if self.args.version in ("v1", "v2") else "diffusion_v2.1"
        self.diffusion: Diffusion = instantiate_from_config(load_config(name))
        self.diffusion.to(self.args.device)

    def load_cond_fn(self) -> None:
        if getattr(self.args, "guidance", False):
            raise NotImplementedError("restoration guidance is dead code in the reference (no sampler calls cond_fn; "
                                      "SURVEY.md §2 #14) and is not implemented")
        self.cond_fn = None

    def load_pipeline(self) -> None:  # pragma: no cover - abstract
        raise NotImplementedError

    def load_captioner(self) -> None:
        if self.args.captioner == "none":
            self.captioner = EmptyCaptioner(self.args.device)
        elif self.args.captioner in ("llava", "ram"):
            raise AssertionError(f"{self.args.captioner} is not available in your environment (captioners are separate "
                                 "products outside this engine; pass --captioner none)")
        else:
            raise ValueError(f"unsupported captioner: {self.args.captioner}")

    def setup(self) -> None:
        self.save_dir = self.args.output
        os.makedirs(self.save_dir, exist_ok=True)

    def load_lq(self) -> Generator["Image.Image", None, None]:  # noqa: F821
        from PIL import Image
        assert os.path.isdir(self.args.input), "Please put your low-quality images in a folder."
        for file_name in sorted(os.listdir(self.args.input)):
            stem, ext = os.path.splitext(file_name)
            if ext not in (".png", ".jpg", ".jpeg"):
                print(f"{file_name} is not an image, continue")
                continue
            file_path = os.path.join(self.args.input, file_name)
            lq = Image.open(file_path).convert("RGB")
            print(f"load lq: {file_path}")
            self.loop_ctx["file_stem"] = stem
            yield lq

    def after_load_lq(self, lq) -> np.ndarray:
        return np.array(lq)

    @torch.no_grad()
    def run(self) -> None:
        """reference loop.py:154-210."""
        self.setup()
        a = self.args
        for lq in self.load_lq():
            caption = self.captioner(lq)
            pos_prompt = ", ".join([text for text in [caption, a.pos_prompt] if text])
            neg_prompt = a.neg_prompt
            lq = self.after_load_lq(lq)
            num_batches = (a.n_samples + a.batch_size - 1) // a.batch_size
            samples = []
            for i in range(num_batches):
                n_inputs = min((i + 1) * a.batch_size, a.n_samples) - i * a.batch_size
                batch_samples = self.pipeline.run(
                    np.tile(lq[None], (n_inputs, 1, 1, 1)), a.steps, a.strength, a.cleaner_tiled, a.cleaner_tile_size,
                    a.cleaner_tile_stride, a.vae_encoder_tiled, a.vae_encoder_tile_size, a.vae_decoder_tiled,
                    a.vae_decoder_tile_size, a.cldm_tiled, a.cldm_tile_size, a.cldm_tile_stride, pos_prompt, neg_prompt,
                    a.cfg_scale, a.start_point_type, a.sampler, a.noise_aug, a.rescale_cfg, a.s_churn, a.s_tmin,
                    a.s_tmax, a.s_noise, a.eta, a.order)
                samples.extend(list(batch_samples))
            self.save(samples, pos_prompt, neg_prompt)

    def save(self, samples: List[np.ndarray], pos_prompt: str, neg_prompt: str) -> None:
        """reference loop.py:212-236: <stem>[_i].png + one prompt.csv row per input image."""
        import csv

        from PIL import Image
        file_stem = self.loop_ctx["file_stem"]
        assert len(samples) == self.args.n_samples
        for i, sample in enumerate(samples):
            file_name = f"{file_stem}_{i}.png" if self.args.n_samples > 1 else f"{file_stem}.png"
            save_path = os.path.join(self.save_dir, file_name)
            Image.fromarray(sample).save(save_path)
            print(f"save result to {save_path}")
        csv_path = os.path.join(self.save_dir, "prompt.csv")
        new = not os.path.exists(csv_path)
        with open(csv_path, "a", newline="") as f:
            w = csv.writer(f)
            if new:
                w.writerow(["file_name", "pos_prompt", "neg_prompt"])
            w.writerow([file_stem, pos_prompt, neg_prompt])
