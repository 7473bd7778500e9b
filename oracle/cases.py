"""ORACLE — TEST INFRASTRUCTURE. Seeded synthetic inputs / weights shared by golden generation and tests."""
import contextlib
import io
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from diffbir_amd import configs
from diffbir_amd.model import specs
from diffbir_amd.utils.synth import synth_state_dict

NEG_PROMPT = "low quality, blurry, low-resolution, noisy, unsharp, weird textures"  # reference inference.py:47-51


def make_lq(seed: int, n: int, h: int, w: int, natural: bool = True) -> np.ndarray:
    """uint8 [n,h,w,3]. natural=True: 1/8-res noise bicubic-upsampled + mild noise."""
    rs = np.random.RandomState(seed)
    if not natural:
        return rs.randint(0, 256, (n, h, w, 3)).astype(np.uint8)
    lo = torch.tensor(rs.rand(n, 3, (h + 7) // 8, (w + 7) // 8), dtype=torch.float32)
    up = F.interpolate(lo, size=(h, w), mode="bicubic", align_corners=False)
    up = up + torch.tensor(rs.randn(n, 3, h, w), dtype=torch.float32) * 0.03
    return (up.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()


class NoiseStream:
    """CPU generator whose draws equal the global CPU RNG after torch.manual_seed(seed) (what the reference
    consumes through torch.randn / randn_like on device='cpu': pipeline.py:159, spaced_sampler.py:181)."""

    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(seed)

    def __call__(self, shape) -> torch.Tensor:
        return torch.randn(tuple(shape), generator=self.g, dtype=torch.float32)


def synth_weights(cldm_cfg: dict, swinir_cfg: dict, seed: int = 0) -> Dict[str, Dict[str, torch.Tensor]]:
    sp = specs.cldm_spec(cldm_cfg)
    W = {k: synth_state_dict(v, seed, prefix=f"{k}.") for k, v in sp.items()}
    W["swinir"] = synth_state_dict(specs.swinir_spec(swinir_cfg), seed, prefix="swinir.")
    return W


# BSRNet / SCUNet cleaners: seeded weights with a gain that keeps activations O(1) through the 23 RRDBs /
# 28 ConvTransBlocks (unit gain grows to 1e5 at the SCUNet output — meaningless in 16 bit), and the module-level inputs
CLEANERS = {
    "bsrnet_tiny": ("TINY_BSRNET", 0.7, (2, 3, 40, 56)), "bsrnet_full": ("FULL_BSRNET", 0.7, (2, 3, 40, 56)),
    "scunet_tiny": ("TINY_SCUNET", 0.6, (2, 3, 100, 136)), "scunet_full": ("FULL_SCUNET", 0.6, (2, 3, 100, 136)),
}


def cleaner_case(name: str):
    """-> (cfg, weights, input f32 NCHW in [0, 1])."""
    key, gain, shape = CLEANERS[name]
    cfg = configs.get(key)
    spec = specs.bsrnet_spec(cfg) if name.startswith("bsrnet") else specs.scunet_spec(cfg)
    W = synth_state_dict(spec, 0, prefix=name.split("_")[0] + ".", gain=gain)
    x = torch.tensor(np.random.RandomState(17).rand(*shape), dtype=torch.float32)
    return cfg, W, x


# end-to-end cases of BSRNetPipeline / SCUNetPipeline on the tiny ControlLDM: name -> (cleaner, lq spec, run kwargs)
CLEANER_PIPELINES = {
    "bsrnet_x4": ("bsrnet_tiny", (21, 1, 128, 128), dict(steps=3, seed=11, upscale=4.0)),
    "bsrnet_x4_tiled": ("bsrnet_tiny", (22, 1, 160, 192), dict(steps=2, seed=11, upscale=4.0, cleaner_tiled=True,
                                                               cleaner_tile=128, cleaner_stride=64)),
    "bsrnet_x2_small": ("bsrnet_tiny", (23, 1, 96, 120), dict(steps=2, seed=11, upscale=2.0)),
    "scunet": ("scunet_tiny", (24, 1, 512, 512), dict(steps=3, seed=11)),
    "scunet_tiled_small": ("scunet_tiny", (25, 1, 320, 384), dict(steps=2, seed=11, cleaner_tiled=True, cleaner_tile=256,
                                                                  cleaner_stride=128)),
}


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def psnr_u8(a: np.ndarray, b: np.ndarray) -> float:
    """PSNR on [0,255] images (same definition as reference utils/common.py:359-390 without Y conversion)."""
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse))


CONFIGS = dict(
    tiny=dict(cldm="TINY_CLDM", swinir="TINY_SWINIR"),
    full=dict(cldm="FULL_CLDM", swinir="FULL_SWINIR"),
)


def get_cfgs(name: str):
    c = CONFIGS[name]
    return configs.get(c["cldm"]), configs.get(c["swinir"])
