"""Blind super-resolution loop (reference diffbir/inference/bsr_loop.py:18-59): SwinIR cleaner for v1 / v2.1, the BSRNet
(RRDBNet x4) cleaner + BSRNetPipeline for v2."""
import numpy as np

from ..pipeline import BSRNetPipeline, SwinIRPipeline
from ..utils.common import instantiate_from_config, load_model_from_url
from .loop import MODELS, InferenceLoop, load_config


class BSRInferenceLoop(InferenceLoop):
    def load_cleaner(self) -> None:
        if self.args.version == "v1":
            config, weight = "swinir", MODELS["swinir_general"]
        elif self.args.version == "v2":
            config, weight = "bsrnet", MODELS["bsrnet"]
        else:
            config, weight = "swinir", MODELS["swinir_realesrgan"]
        self.cleaner = instantiate_from_config(load_config(config))
        self.cleaner.load_state_dict(load_model_from_url(weight), strict=True)
        self.cleaner.eval().to(self.args.device)

    def load_pipeline(self) -> None:
        if self.args.version in ("v1", "v2.1"):
            self.pipeline = SwinIRPipeline(self.cleaner, self.cldm, self.diffusion, self.cond_fn, self.args.device)
        else:
            self.pipeline = BSRNetPipeline(self.cleaner, self.cldm, self.diffusion, self.cond_fn, self.args.device,
                                           self.args.upscale)

    def after_load_lq(self, lq) -> np.ndarray:
        if self.args.version in ("v1", "v2.1"):   # BSRNet up-scales by itself (bsr_loop.py:54-58)
            from PIL import Image
            lq = lq.resize(tuple(int(x * self.args.upscale) for x in lq.size), Image.BICUBIC)
        return super().after_load_lq(lq)
