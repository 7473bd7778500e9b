"""Restoration pipeline with the reference's call surface (reference diffbir/pipeline.py:47-321, 369-397).

`SwinIRPipeline(cleaner, cldm, diffusion, cond_fn, device).run(lq, steps, strength, <tiling args>, pos_prompt,
neg_prompt, cfg_scale, start_point_type, sampler_type, noise_aug, rescale_cfg, s_churn, s_tmin, s_tmax, s_noise,
eta, order) -> np.uint8[N,H,W,3]` — same positional signature, same errors.
uint8 -> f32, the wavelet colour fix and f32 -> uint8 run as HIP kernels; bicubic-antialias resizes and the zero /
reflect pads only trigger for sizes that are not multiples of 64 and stay on PyTorch (cold).
"""
from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .sampler import DDIMSampler, DPMSolverSampler, EDMSampler, SpacedSampler
from .utils.common import make_tiled_fn, wavelet_reconstruction


def resize_short_edge_to(imgs: torch.Tensor, size: int) -> torch.Tensor:
    """reference pipeline.py:25-34."""
    _, _, h, w = imgs.size()
    if h == w:
        out_h, out_w = size, size
    elif h < w:
        out_h, out_w = size, int(w * (size / h))
    else:
        out_h, out_w = int(h * (size / w)), size
    return F.interpolate(imgs, size=(out_h, out_w), mode="bicubic", antialias=True)


def pad_to_multiples_of(imgs: torch.Tensor, multiple: int) -> torch.Tensor:
    """reference pipeline.py:37-42 (zero pad right / bottom)."""
    _, _, h, w = imgs.size()
    if h % multiple == 0 and w % multiple == 0:
        return imgs.clone()
    ph, pw = ((x + multiple - 1) // multiple * multiple - x for x in (h, w))
    return F.pad(imgs, pad=(0, pw, 0, ph), mode="constant", value=0)


class Pipeline:
    def __init__(self, cleaner, cldm, diffusion, cond_fn, device: str) -> None:
        self.cleaner = cleaner
        self.cldm = cldm
        self.diffusion = diffusion
        self.cond_fn = cond_fn
        self.device = device
        self.output_size: Tuple[int, int] = None
        if cond_fn is not None:
            raise NotImplementedError("restoration guidance is dead code in the reference (no sampler calls cond_fn; "
                                      "SURVEY.md §2 #14) and is not implemented")
        # engine extension (tests / data-parallel sharding): Gaussian noise source, shape -> f32 tensor.
        self.randn: Optional[Callable] = None
        # engine extension (diffbir_amd.parallel): shard the tiles of tiled sampling over (rank, world)
        self.tile_shard: Optional[Tuple[int, int]] = None
        self.tile_all_reduce: Optional[Callable] = None
        # engine extension: Brownian-motion factory handed to the EDM SDE solvers (sampler/edm_sampler.py; None = the restated torchsde tree)
        self.brownian: Optional[Callable] = None

    def _randn(self, shape) -> torch.Tensor:
        if self.randn is not None:
            return self.randn(tuple(shape)).to(device=self.device, dtype=torch.float32).contiguous()
        return torch.randn(tuple(shape), dtype=torch.float32, device=self.device)

    def set_output_size(self, lq_size: Tuple[int]) -> None:
        h, w = lq_size[2:]
        self.output_size = (h, w)

    def apply_cleaner(self, lq, tiled, tile_size, tile_stride):  # pragma: no cover - abstract
        raise NotImplementedError

    def apply_cldm(self, cond_img, steps, strength, vae_encoder_tiled, vae_encoder_tile_size, vae_decoder_tiled,
                   vae_decoder_tile_size, cldm_tiled, cldm_tile_size, cldm_tile_stride, pos_prompt, neg_prompt,
                   cfg_scale, start_point_type, sampler_type, noise_aug, rescale_cfg, s_churn, s_tmin, s_tmax,
                   s_noise, eta, order) -> torch.Tensor:
        """reference pipeline.py:71-233."""
        bs, _, h0, w0 = cond_img.shape
        if not vae_encoder_tiled and not cldm_tiled:
            cond_img = pad_to_multiples_of(cond_img, multiple=64)
        else:
            cond_img = pad_to_multiples_of(cond_img, multiple=8)
        if vae_encoder_tiled and (cond_img.size(2) < vae_encoder_tile_size or cond_img.size(3) < vae_encoder_tile_size):
            print("[VAE Encoder]: the input size is tiny and unnecessary to tile.")
            vae_encoder_tiled = False
        if vae_encoder_tiled and vae_encoder_tile_size % 8 != 0:
            raise ValueError("VAE encoder tile size must be a multiple of 8")
        # the reference encodes the same condition image twice (pos / neg prompt, pipeline.py:117-128); mode() is
        # deterministic, so encode once and share the latent (bit-identical result)
        cond = self.cldm.prepare_condition(cond_img, [pos_prompt] * bs, vae_encoder_tiled, vae_encoder_tile_size)
        uncond = dict(c_txt=self.cldm.clip.encode([neg_prompt] * bs), c_img=cond["c_img"].clone())
        h1, w1 = cond["c_img"].shape[2:]
        if cldm_tiled and (h1 < cldm_tile_size // 8 or w1 < cldm_tile_size // 8):
            print("[Diffusion]: the input size is tiny and unnecessary to tile.")
            cldm_tiled = False
        if not cldm_tiled:
            cond["c_img"] = pad_to_multiples_of(cond["c_img"], multiple=8)
            uncond["c_img"] = pad_to_multiples_of(uncond["c_img"], multiple=8)
        elif cldm_tile_size % 64 != 0:
            raise ValueError("Diffusion tile size must be a multiple of 64")
        h2, w2 = cond["c_img"].shape[2:]
        if start_point_type == "cond":
            x_0 = cond["c_img"]
            t_last = torch.full((bs,), self.diffusion.num_timesteps - 1, dtype=torch.long, device=self.device)
            x_T = self.diffusion.q_sample(x_0, t_last, self._randn(x_0.shape))
        else:
            x_T = self._randn((bs, 4, h2, w2))
        if noise_aug > 0:
            cond["c_img"] = self.diffusion.q_sample(
                cond["c_img"], torch.full((bs,), noise_aug, dtype=torch.long, device=self.device),
                self._randn(cond["c_img"].shape))
            uncond["c_img"] = cond["c_img"].detach().clone()
        control_scales = self.cldm.control_scales
        self.cldm.control_scales = [strength] * 13
        betas, parameterization = self.diffusion.betas, self.diffusion.parameterization
        if sampler_type == "spaced":
            sampler = SpacedSampler(betas, parameterization, rescale_cfg)
        elif sampler_type == "ddim":
            sampler = DDIMSampler(betas, parameterization, rescale_cfg, eta=0)
        elif sampler_type.startswith("dpm"):
            sampler = DPMSolverSampler(betas, parameterization, rescale_cfg, sampler_type)
        elif sampler_type.startswith("edm"):
            sampler = EDMSampler(betas, parameterization, rescale_cfg, sampler_type, s_churn, s_tmin, s_tmax, s_noise,
                                 eta, order)
        else:
            raise NotImplementedError(sampler_type)
        sampler.randn = self.randn
        if self.brownian is not None and hasattr(sampler, "brownian"):
            sampler.brownian = self.brownian
        sampler.tile_shard, sampler.tile_all_reduce = self.tile_shard, self.tile_all_reduce
        try:
            z = sampler.sample(model=self.cldm, device=self.device, steps=steps, x_size=(bs, 4, h2, w2), cond=cond,
                               uncond=uncond, cfg_scale=cfg_scale, tiled=cldm_tiled, tile_size=cldm_tile_size // 8,
                               tile_stride=cldm_tile_stride // 8, x_T=x_T, progress=False)
        finally:
            self.cldm.control_scales = control_scales
        z = z[..., :h1, :w1].contiguous()
        if vae_decoder_tiled and (h1 < vae_decoder_tile_size // 8 or w1 < vae_decoder_tile_size // 8):
            print("[VAE Decoder]: the input size is tiny and unnecessary to tile.")
            vae_decoder_tiled = False
        x = self.cldm.vae_decode(z, vae_decoder_tiled, vae_decoder_tile_size // 8)
        return x[:, :, :h0, :w0]

    @torch.no_grad()
    def run(self, lq: np.ndarray, steps: int, strength: float, cleaner_tiled: bool, cleaner_tile_size: int,
            cleaner_tile_stride: int, vae_encoder_tiled: bool, vae_encoder_tile_size: int, vae_decoder_tiled: bool,
            vae_decoder_tile_size: int, cldm_tiled: bool, cldm_tile_size: int, cldm_tile_stride: int, pos_prompt: str,
            neg_prompt: str, cfg_scale: float, start_point_type: str, sampler_type: str, noise_aug: int,
            rescale_cfg: bool, s_churn: float, s_tmin: float, s_tmax: float, s_noise: float, eta: float,
            order: int) -> np.ndarray:
        """reference pipeline.py:235-321."""
        if isinstance(lq, torch.Tensor):  # engine extension: batch already resident in HBM
            lq_u8 = lq.to(device=self.device, dtype=torch.uint8).contiguous()
        else:
            lq_u8 = torch.as_tensor(np.ascontiguousarray(lq), dtype=torch.uint8).to(self.device)
        lq_tensor = ops.u8_to_f32_nchw(lq_u8)
        self.set_output_size(lq_tensor.size())
        cond_img = self.apply_cleaner(lq_tensor, cleaner_tiled, cleaner_tile_size, cleaner_tile_stride)
        assert all(x >= 512 for x in cond_img.shape[2:]), (
            "The resolution of stage-1 model output should be greater than 512, "
            "since it will be used as condition for stage-2 model.")
        sample = self.apply_cldm(cond_img, steps, strength, vae_encoder_tiled, vae_encoder_tile_size,
                                 vae_decoder_tiled, vae_decoder_tile_size, cldm_tiled, cldm_tile_size,
                                 cldm_tile_stride, pos_prompt, neg_prompt, cfg_scale, start_point_type, sampler_type,
                                 noise_aug, rescale_cfg, s_churn, s_tmin, s_tmax, s_noise, eta, order)
        sample = wavelet_reconstruction(((sample + 1) / 2).contiguous(), cond_img.contiguous())
        if tuple(sample.shape[2:]) != tuple(self.output_size):
            # bicubic+antialias is an exact no-op when sizes match; otherwise cold path on PyTorch
            sample = F.interpolate(sample, size=self.output_size, mode="bicubic", antialias=True)
        return ops.f32_nchw_to_u8_nhwc(sample.contiguous()).cpu().numpy()


class SwinIRPipeline(Pipeline):
    def apply_cleaner(self, lq: torch.Tensor, tiled: bool, tile_size: int, tile_stride: int) -> torch.Tensor:
        """reference pipeline.py:371-397."""
        if tiled and (lq.size(2) < tile_size or lq.size(3) < tile_size):
            print("[SwinIR]: the input size is tiny and unnecessary to tile.")
            tiled = False
        if tiled and tile_size % 64 != 0:
            raise ValueError("SwinIR (cleaner) tile size must be a multiple of 64")
        if not tiled:
            if min(lq.shape[2:]) < 512:
                lq = resize_short_edge_to(lq, size=512)
            h0, w0 = lq.shape[2:]
            lq = pad_to_multiples_of(lq, multiple=64)
            return self.cleaner(lq)[:, :, :h0, :w0]
        output = make_tiled_fn(self.cleaner, size=tile_size, stride=tile_stride)(lq)
        if min(output.shape[2:]) < 512:
            output = resize_short_edge_to(output, size=512)
        return output


class BSRNetPipeline(Pipeline):
    """reference pipeline.py:324-366: the x4 RRDBNet cleaner runs on the LQ image itself; its output is resized to the
    requested upscale (bicubic + antialias: cold path on PyTorch, like the reference)."""

    def __init__(self, cleaner, cldm, diffusion, cond_fn, device: str, upscale: float) -> None:
        super().__init__(cleaner, cldm, diffusion, cond_fn, device)
        self.upscale = upscale

    def set_output_size(self, lq_size: Tuple[int]) -> None:
        h, w = lq_size[2:]
        self.output_size = (int(h * self.upscale), int(w * self.upscale))

    def apply_cleaner(self, lq: torch.Tensor, tiled: bool, tile_size: int, tile_stride: int) -> torch.Tensor:
        if tiled and (lq.size(2) < tile_size or lq.size(3) < tile_size):
            print("[BSRNet]: the input size is tiny and unnecessary to tile.")
            tiled = False
        model = make_tiled_fn(self.cleaner, tile_size, tile_stride, scale_type="up", scale=4) if tiled else self.cleaner
        output_upscale4 = model(lq)
        if min(self.output_size) < 512:
            return resize_short_edge_to(output_upscale4, size=512)
        return F.interpolate(output_upscale4, size=self.output_size, mode="bicubic", antialias=True)


class SCUNetPipeline(Pipeline):
    """reference pipeline.py:400-420 (blind image denoising, DiffBIR v2)."""

    def apply_cleaner(self, lq: torch.Tensor, tiled: bool, tile_size: int, tile_stride: int) -> torch.Tensor:
        if tiled and (lq.size(2) < tile_size or lq.size(3) < tile_size):
            print("[SCUNet]: the input size is tiny and unnecessary to tile.")
            tiled = False
        model = make_tiled_fn(self.cleaner, tile_size, tile_stride) if tiled else self.cleaner
        output = model(lq)
        if min(output.shape[2:]) < 512:
            output = resize_short_edge_to(output, size=512)
        return output
