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

    def _randn(self, shape, device) -> torch.Tensor:
        if self.randn is not None:
            return self.randn(tuple(shape)).to(device=device, dtype=torch.float32).contiguous()
        return torch.randn(tuple(shape), device=device, dtype=torch.float32)

    def get_cfg_scale(self, default_cfg_scale: float, model_t: int) -> float:
        """reference sampler.py:31-38."""
        if self.rescale_cfg and default_cfg_scale > 1:
            return 1 + default_cfg_scale * ((1 - math.cos(math.pi * ((1000 - model_t) / 1000) ** 5.0)) / 2)
        return default_cfg_scale
