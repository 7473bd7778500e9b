"""Blind image denoising loop (reference diffbir/inference/bid_loop.py:18-58): SwinIR cleaner for v1 / v2.1, the SCUNet
cleaner + SCUNetPipeline for v2."""
import numpy as np

from ..pipeline import SCUNetPipeline, SwinIRPipeline
from ..utils.common import instantiate_from_config, load_model_from_url
from .loop import MODELS, InferenceLoop, load_config


class BIDInferenceLoop(InferenceLoop):
    def load_cleaner(self) -> None:
        if self.args.version == "v1":
            config, weight = "swinir", MODELS["swinir_general"]
        elif self.args.version == "v2":
            config, weight = "scunet", MODELS["scunet_psnr"]
        else:
            config, weight = "swinir", MODELS["swinir_realesrgan"]
        self.cleaner = instantiate_from_config(load_config(config))
        self.cleaner.load_state_dict(load_model_from_url(weight), strict=True)
        self.cleaner.eval().to(self.args.device)

    def load_pipeline(self) -> None:
        cls = SwinIRPipeline if self.args.version in ("v1", "v2.1") else SCUNetPipeline
        self.pipeline = cls(self.cleaner, self.cldm, self.diffusion, self.cond_fn, self.args.device)

    def after_load_lq(self, lq) -> np.ndarray:
        from PIL import Image
        lq = lq.resize(tuple(int(x * self.args.upscale) for x in lq.size), Image.BICUBIC)
        return super().after_load_lq(lq)
