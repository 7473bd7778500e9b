"""Sampler base (reference sampler/sampler.py:10-55)."""
import math
from typing import Callable, Optional

import numpy as np
import torch


class Sampler:
    def __init__(self, betas: np.ndarray, parameterization: str, rescale_cfg: bool):
        self.num_timesteps = len(betas)
        self.training_betas = betas
        self.training_alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        self.context = {}
        self.parameterization = parameterization
        self.rescale_cfg = rescale_cfg
        # engine extension: source of Gaussian noise (shape -> f32 tensor on the sampling device). Default is the
        # device generator, exactly what the reference consumes (torch.randn / randn_like on `device`).
        self.randn: Optional[Callable] = None
        # engine extension (diffbir_amd.parallel): (rank, world) tile shard + all-reduce callable for tiled sampling
        self.tile_shard = None
        self.tile_all_reduce: Optional[Callable] = None

    def _tiled(self, forward: Callable, tile_size: int, tile_stride: int):
        from ..utils.tiling import TiledModel
        return TiledModel(forward, tile_size, tile_stride, shard=self.tile_shard, all_reduce=self.tile_all_reduce)

    @staticmethod
    def _cfg_batch(cond, uncond, bs: int) -> dict:
        """The [uncond || cond] conditioning of the ONE batch-2B network evaluation that replaces the reference's two
        batch-B forwards per step (spaced_sampler.py:156-157).  When both halves carry the same condition latent (they
        always do when built by Pipeline.apply_cldm, pipeline.py:117-128) the batch is flagged `cfg_pair` = (groups,
        bs): x, t and c_img of the two halves are identical, so the networks may share what precedes their first
        cross-attention (model/unet.py).  Checked once per sample() call, not per step."""
        cond2 = {k: torch.cat([uncond[k], cond[k]], dim=0).contiguous() for k in ("c_txt", "c_img")}
        if uncond["c_img"].shape == cond["c_img"].shape and torch.equal(uncond["c_img"], cond["c_img"]):
            cond2["cfg_pair"] = (1, bs)
        return cond2

    def _randn(self, shape, device) -> torch.Tensor:
        if self.randn is not None:
            return self.randn(tuple(shape)).to(device=device, dtype=torch.float32).contiguous()
        return torch.randn(tuple(shape), device=device, dtype=torch.float32)

    def get_cfg_scale(self, default_cfg_scale: float, model_t: int) -> float:
        """reference sampler.py:31-38."""
        if self.rescale_cfg and default_cfg_scale > 1:
            return 1 + default_cfg_scale * ((1 - math.cos(math.pi * ((1000 - model_t) / 1000) ** 5.0)) / 2)
        return default_cfg_scale
