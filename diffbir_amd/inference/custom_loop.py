"""Inference with self-trained weights (reference diffbir/inference/custom_loop.py:20-106): `--version custom --train_cfg
<stage-2 training YAML> --ckpt <ControlNet checkpoint>`.  The training config supplies the `model.cldm` / `model.swinir` /
`model.diffusion` trees and the paths of the pre-trained SD and SwinIR checkpoints (`train.sd_path`, `train.swinir_path`)."""
from argparse import Namespace

import numpy as np
import torch

from ..pipeline import SwinIRPipeline
from ..utils.common import instantiate_from_config
from .loop import InferenceLoop


def _load_yaml(path: str) -> dict:
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


class CustomInferenceLoop(InferenceLoop):
    def __init__(self, args: Namespace) -> None:
        self.train_cfg = _load_yaml(args.train_cfg)
        super().__init__(args)

    def load_cleaner(self) -> None:
        """The stage-1 model of the training config (SwinIR), its checkpoint from `train.swinir_path` (state_dict
        unwrapped, `module.` prefixes stripped)."""
        self.cleaner = instantiate_from_config(self.train_cfg["model"]["swinir"])
        weight = torch.load(self.train_cfg["train"]["swinir_path"], map_location="cpu")
        if "state_dict" in weight:
            weight = weight["state_dict"]
        weight = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in weight.items()}
        self.cleaner.load_state_dict(weight, strict=True)
        self.cleaner.eval().to(self.args.device)

    def load_cldm(self) -> None:
        self.cldm = instantiate_from_config(self.train_cfg["model"]["cldm"])
        sd_weight = torch.load(self.train_cfg["train"]["sd_path"], map_location="cpu")["state_dict"]
        unused, missing = self.cldm.load_pretrained_sd(sd_weight)
        print(f"load pretrained stable diffusion, unused weights: {unused}, missing weights: {missing}")
        self.cldm.load_controlnet_from_ckpt(torch.load(self.args.ckpt, map_location="cpu"))
        print("load controlnet weight")
        self.cldm.eval().to(self.args.device)
        cast_type = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[self.args.precision]
        self.cldm.cast_dtype(cast_type)
        self.diffusion = instantiate_from_config(self.train_cfg["model"]["diffusion"])
        self.diffusion.to(self.args.device)

    def load_pipeline(self) -> None:
        self.pipeline = SwinIRPipeline(self.cleaner, self.cldm, self.diffusion, self.cond_fn, self.args.device)

    def after_load_lq(self, lq) -> np.ndarray:
        from PIL import Image
        lq = lq.resize(tuple(int(x * self.args.upscale) for x in lq.size), Image.BICUBIC)
        return super().after_load_lq(lq)
