"""Blind face restoration loop for aligned faces (reference diffbir/inference/bfr_loop.py:15-34)."""
import numpy as np

from ..pipeline import SwinIRPipeline
from ..utils.common import instantiate_from_config, load_model_from_url
from .loop import MODELS, InferenceLoop, load_config


class BFRInferenceLoop(InferenceLoop):
    def load_cleaner(self) -> None:
        self.cleaner = instantiate_from_config(load_config("swinir"))
        self.cleaner.load_state_dict(load_model_from_url(MODELS["swinir_face"]), strict=True)
        self.cleaner.eval().to(self.args.device)

    def load_pipeline(self) -> None:
        self.pipeline = SwinIRPipeline(self.cleaner, self.cldm, self.diffusion, self.cond_fn, self.args.device)

    def after_load_lq(self, lq) -> np.ndarray:
        from PIL import Image
        lq = lq.resize(tuple(int(x * self.args.upscale) for x in lq.size), Image.BICUBIC)
        return super().after_load_lq(lq)
