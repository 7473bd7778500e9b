"""Blind super-resolution loop (reference diffbir/inference/bsr_loop.py:18-59)."""
import numpy as np

from ..pipeline import SwinIRPipeline
from ..utils.common import instantiate_from_config, load_model_from_url
from .loop import MODELS, InferenceLoop, load_config


class BSRInferenceLoop(InferenceLoop):
    def load_cleaner(self) -> None:
        if self.args.version == "v2":
            raise NotImplementedError("DiffBIR v2 BSR uses the BSRNet (RRDBNet) cleaner, which is outside this engine's "
                                      "hot-path scope (SURVEY.md §8f N3); use --version v1 or v2.1 (SwinIR)")
        weight = MODELS["swinir_general" if self.args.version == "v1" else "swinir_realesrgan"]
        self.cleaner = instantiate_from_config(load_config("swinir"))
        self.cleaner.load_state_dict(load_model_from_url(weight), strict=True)
        self.cleaner.eval().to(self.args.device)

    def load_pipeline(self) -> None:
        self.pipeline = SwinIRPipeline(self.cleaner, self.cldm, self.diffusion, self.cond_fn, self.args.device)

    def after_load_lq(self, lq) -> np.ndarray:
        from PIL import Image
        lq = lq.resize(tuple(int(x * self.args.upscale) for x in lq.size), Image.BICUBIC)
        return super().after_load_lq(lq)
