"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU / fp32 restatement of `SwinIRPipeline.run` (reference pipeline.py:235-321, 371-397, 71-233) built on
oracle.nets / oracle.sampling.  Also used (bounded sample) as the `cpu_baseline` ("port") in bench.py.
"""
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import nets, sampling

T = torch.Tensor


def wavelet_blur(img: T, radius: int) -> T:
    """reference utils/common.py:29-47."""
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]],
                     dtype=img.dtype)[None, None].repeat(3, 1, 1, 1)
    img = F.pad(img, (radius,) * 4, mode="replicate")
    return F.conv2d(img, k, groups=3, dilation=radius)


def wavelet_reconstruction(content: T, style: T, levels: int = 5) -> T:
    """reference utils/common.py:50-77: high-freq of content + low-freq of style."""

    def decomp(im):
        high = torch.zeros_like(im)
        for i in range(levels):
            low = wavelet_blur(im, 2 ** i)
            high = high + (im - low)
            im = low
        return high, low

    ch, _ = decomp(content)
    _, sl = decomp(style)
    return ch + sl


def pad_to_multiple(x: T, m: int) -> T:
    """reference pipeline.py:37-42 (zero pad right/bottom)."""
    h, w = x.shape[2:]
    ph, pw = (-h) % m, (-w) % m
    return F.pad(x, (0, pw, 0, ph)) if (ph or pw) else x.clone()


def resize_short_edge_to(x: T, size: int) -> T:
    """reference pipeline.py:25-34."""
    h, w = x.shape[2:]
    if h == w:
        oh, ow = size, size
    elif h < w:
        oh, ow = size, int(w * (size / h))
    else:
        oh, ow = int(h * (size / w)), size
    return F.interpolate(x, size=(oh, ow), mode="bicubic", antialias=True)


class OraclePipeline:
    """Mirror of SwinIRPipeline with functional nets. ``W`` = dict(unet=, controlnet=, vae=, clip=, swinir=)."""

    def __init__(self, W: Dict[str, Dict[str, T]], cldm_cfg: dict, swinir_cfg: dict, diffusion_cfg: dict,
                 tokenize: Callable[[List[str]], T]):
        self.W, self.cldm_cfg, self.swinir_cfg = W, cldm_cfg, swinir_cfg
        self.betas = sampling.make_betas(**diffusion_cfg)
        self.parameterization = diffusion_cfg.get("parameterization", "eps")
        self.scale_factor = cldm_cfg["latent_scale_factor"]
        self.tokenize = tokenize
        self.control_scales = [1.0] * 13

    # --- ControlLDM surface (reference cldm.py) ---
    def model(self, x: T, t: T, cond: dict) -> T:
        return nets.cldm_forward(self.W, self.cldm_cfg, x, t, cond["c_txt"], cond["c_img"], self.control_scales)

    def prepare_condition(self, img: T, txt: List[str]) -> dict:
        return dict(
            c_txt=nets.clip_text_encode(self.W["clip"], self.cldm_cfg["clip_cfg"], self.tokenize(txt)),
            c_img=nets.vae_encode_mode(self.W["vae"], self.cldm_cfg["vae_cfg"], img * 2 - 1, self.scale_factor))

    def vae_decode(self, z: T) -> T:
        return nets.vae_decode(self.W["vae"], self.cldm_cfg["vae_cfg"], z, self.scale_factor)

    def cleaner(self, x: T) -> T:
        return nets.swinir_forward(self.W["swinir"], self.swinir_cfg, x)

    def apply_cleaner(self, lq: T, tiled=False, tile_size=512, tile_stride=256) -> T:
        """reference pipeline.py:371-397."""
        if tiled and (lq.shape[2] < tile_size or lq.shape[3] < tile_size):
            tiled = False
        if tiled and tile_size % 64 != 0:
            raise ValueError("SwinIR (cleaner) tile size must be a multiple of 64")
        if not tiled:
            if min(lq.shape[2:]) < 512:
                lq = resize_short_edge_to(lq, 512)
            h0, w0 = lq.shape[2:]
            return self.cleaner(pad_to_multiple(lq, 64))[:, :, :h0, :w0]
        out = sampling.make_tiled_fn(self.cleaner, tile_size, tile_stride)(lq)
        if min(out.shape[2:]) < 512:
            out = resize_short_edge_to(out, 512)
        return out

    def apply_cldm(self, cond_img: T, steps: int, strength: float, cldm_tiled: bool, tile_size: int,
                   tile_stride: int, pos_prompt: str, neg_prompt: str, cfg_scale: float, start_point_type: str,
                   sampler_type: str, noise_aug: int, rescale_cfg: bool, randn: Callable, taps: dict = None) -> T:
        """reference pipeline.py:71-233 (vae tiling flags off)."""
        bs, _, h0, w0 = cond_img.shape
        cond_img = pad_to_multiple(cond_img, 8 if cldm_tiled else 64)
        cond = self.prepare_condition(cond_img, [pos_prompt] * bs)
        uncond = self.prepare_condition(cond_img, [neg_prompt] * bs)
        h1, w1 = cond["c_img"].shape[2:]
        if cldm_tiled and (h1 < tile_size // 8 or w1 < tile_size // 8):
            cldm_tiled = False
        if not cldm_tiled:
            cond["c_img"] = pad_to_multiple(cond["c_img"], 8)
            uncond["c_img"] = pad_to_multiple(uncond["c_img"], 8)
        elif tile_size % 64 != 0:
            raise ValueError("Diffusion tile size must be a multiple of 64")
        h2, w2 = cond["c_img"].shape[2:]
        if start_point_type == "cond":
            t_last = torch.full((bs,), len(self.betas) - 1, dtype=torch.long)
            x_T = sampling.q_sample(self.betas, cond["c_img"], t_last, randn(tuple(cond["c_img"].shape)))
        else:
            x_T = randn((bs, 4, h2, w2))
        if noise_aug > 0:
            cond["c_img"] = sampling.q_sample(self.betas, cond["c_img"], torch.full((bs,), noise_aug),
                                              randn(tuple(cond["c_img"].shape)))
            uncond["c_img"] = cond["c_img"].clone()
        if taps is not None:
            taps.update(c_img=cond["c_img"].clone(), c_txt=cond["c_txt"].clone(), x_T=x_T.clone())
        saved = self.control_scales
        self.control_scales = [strength] * 13
        kw = dict(tiled=cldm_tiled, tile_size=tile_size // 8, tile_stride=tile_stride // 8)
        if sampler_type == "spaced":
            z = sampling.spaced_sample(self.model, self.betas, self.parameterization, steps, x_T, cond, uncond,
                                       cfg_scale, rescale_cfg, noise_fn=lambda x: randn(tuple(x.shape)), **kw)
        elif sampler_type.startswith("dpm++_m"):
            z = sampling.dpmpp_2m_sample(self.model, self.betas, self.parameterization, steps, x_T, cond, uncond,
                                         cfg_scale, order=int(sampler_type[-1]), **kw)
        else:
            raise NotImplementedError(sampler_type)
        z = z[..., :h1, :w1]
        if taps is not None:
            taps.update(z=z.clone())
        x = self.vae_decode(z)
        self.control_scales = saved
        return x[:, :, :h0, :w0]

    @torch.no_grad()
    def run(self, lq: np.ndarray, steps: int, strength: float = 1.0, cleaner_tiled=False, cleaner_tile_size=512,
            cleaner_tile_stride=256, cldm_tiled=False, cldm_tile_size=512, cldm_tile_stride=256,
            pos_prompt="", neg_prompt="", cfg_scale=4.0, start_point_type="noise", sampler_type="spaced",
            noise_aug=0, rescale_cfg=False, randn: Optional[Callable] = None, taps: dict = None) -> np.ndarray:
        """reference pipeline.py:235-321."""
        randn = randn or (lambda shape: torch.randn(shape, dtype=torch.float32))
        x = torch.tensor(lq, dtype=torch.float32).div(255).clamp(0, 1).permute(0, 3, 1, 2).contiguous()
        out_size = tuple(x.shape[2:])
        cond_img = self.apply_cleaner(x, cleaner_tiled, cleaner_tile_size, cleaner_tile_stride)
        assert all(s >= 512 for s in cond_img.shape[2:])
        if taps is not None:
            taps.update(clean=cond_img.clone())
        sample = self.apply_cldm(cond_img, steps, strength, cldm_tiled, cldm_tile_size, cldm_tile_stride,
                                 pos_prompt, neg_prompt, cfg_scale, start_point_type, sampler_type, noise_aug,
                                 rescale_cfg, randn, taps)
        if taps is not None:
            taps.update(decoded=sample.clone())
        sample = F.interpolate(wavelet_reconstruction((sample + 1) / 2, cond_img), size=out_size,
                               mode="bicubic", antialias=True)
        return (sample * 255.0).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()
