"""DDIM sampler (reference sampler/ddim_sampler.py:13-203) on the engine.

Host tables are the reference's (float64 numpy, registered as f32).  Per step the engine runs ONE batched network
evaluation (uncond || cond at batch 2B — the reference cats [cond, uncond]; identical per sample) and ONE fused f32
kernel for CFG mix + eps / x0 conversion + the DDIM update: with A = sqrt(a_t), S = sqrt(1 - a_t), P = sqrt(a_prev),
D = sqrt(1 - a_prev - sigma_t^2) and the CFG mix m = s*oc + (1 - s)*ou, the reference's

    e = m (eps)  or  A*m + S*x (v);   x0 = (x - S*e) / A;   x_prev = P*x0 + D*e + sigma_t * noise

is the linear combination  x_prev = kx*x + km*m + sigma_t*noise  with
    eps:  kx = P / A,        km = D - P*S / A
    v:    kx = P*A + D*S,    km = D*A - P*S
(`dbir_lincomb4`).  One Gaussian draw per step even when sigma_t = 0 (the reference calls randn_like unconditionally,
ddim_sampler.py:143: the device RNG stream stays aligned).
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import ops
from .sampler import Sampler


def make_ddim_timesteps(method: str, num_ddim: int, num_ddpm: int) -> np.ndarray:
    """reference ddim_sampler.py:13-36 (`+ 1`: the final alpha values come out right)."""
    if method == "uniform":
        steps = np.asarray(list(range(0, num_ddpm, num_ddpm // num_ddim)))
    elif method == "quad":
        steps = ((np.linspace(0, np.sqrt(num_ddpm * 0.8), num_ddim)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{method}"')
    return steps + 1


def make_ddim_sampling_parameters(alphacums: np.ndarray, ddim_timesteps: np.ndarray, eta: float):
    """reference ddim_sampler.py:39-58 (Song et al. 2020, eq. 16)."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


class DDIMSampler(Sampler):
    def __init__(self, betas: np.ndarray, parameterization: str, rescale_cfg: bool, eta: float):
        super().__init__(betas, parameterization, rescale_cfg)
        self.eta = eta

    def make_schedule(self, ddim_num_steps: int, ddim_discretize: str = "uniform") -> None:
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.num_timesteps)
        sig, a, ap = make_ddim_sampling_parameters(self.training_alphas_cumprod, self.ddim_timesteps, self.eta)
        f32 = lambda v: np.asarray(v, dtype=np.float64).astype(np.float32)   # `register` stores f32 buffers
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = f32(sig), f32(a), f32(ap)
        self.ddim_sqrt_alphas, self.ddim_sqrt_one_minus_alphas = f32(np.sqrt(a)), f32(np.sqrt(1.0 - a))

    @torch.no_grad()
    def sample(self, model, device: str, steps: int, x_size: Tuple[int], cond: Dict[str, torch.Tensor],
               uncond: Optional[Dict[str, torch.Tensor]], cfg_scale: float, tiled: bool = False, tile_size: int = -1,
               tile_stride: int = -1, x_T: Optional[torch.Tensor] = None, progress: bool = True) -> torch.Tensor:
        self.make_schedule(ddim_num_steps=steps)
        bs = x_size[0]
        fwd = model.forward
        if tiled:
            fwd = self._tiled(model.forward, tile_size, tile_stride)
        if x_T is None:
            x_T = self._randn(x_size, device)
        x = x_T.to(device=device, dtype=torch.float32).contiguous()
        use_cfg = not (uncond is None or cfg_scale == 1.0)
        if use_cfg:
            cond2 = self._cfg_batch(cond, uncond, bs)
        total = self.ddim_timesteps.shape[0]
        it = list(enumerate(np.flip(self.ddim_timesteps)))
        if progress:
            from tqdm import tqdm
            it = tqdm(it, desc="DDIM Sampler", total=total)
        full = lambda v: torch.full((bs,), float(v), device=device, dtype=torch.float32)
        for i, step in it:
            ti = total - i - 1
            model_t = torch.full((bs,), int(step), device=device, dtype=torch.float32)
            s = float(self.get_cfg_scale(cfg_scale, int(step)))
            if use_cfg:   # decided once from the caller's cfg_scale; the reference decides per step from the rescaled
                          # scale (ddim_sampler.py:150-160) — identical results: where that scale is exactly 1 the
                          # unconditional term below is multiplied by 0
                o = fwd(torch.cat([x, x], dim=0), torch.cat([model_t, model_t]), cond2)
                ou, oc = o[:bs].contiguous(), o[bs:].contiguous()
            else:
                oc, ou, s = fwd(x, model_t, cond).contiguous(), None, 1.0
            # f32 scalars, evaluated like the reference's tensor expressions
            a_t, a_prev = np.float32(self.ddim_alphas[ti]), np.float32(self.ddim_alphas_prev[ti])
            sigma, S = np.float32(self.ddim_sigmas[ti]), np.float32(self.ddim_sqrt_one_minus_alphas[ti])
            A_tab = np.float32(self.ddim_sqrt_alphas[ti])
            A, P = np.sqrt(a_t), np.sqrt(a_prev)
            D = np.sqrt(np.float32(1.0) - a_prev - sigma * sigma)
            if self.parameterization == "eps":
                kx, km = P / A, D - P * S / A
            else:   # e = A_tab*v + S*x (predict_eps_from_z_and_v), then x0 = (x - S*e) / A
                kx, km = P * (np.float32(1.0) - S * S) / A + D * S, D * A_tab - P * S * A_tab / A
            noise = self._randn(x.shape, device)
            if ou is None:
                x = ops.lincomb4(x, full(kx), oc, full(km), noise, full(sigma))
            else:
                x = ops.lincomb4(x, full(kx), oc, full(km * s), ou, full(km * (1.0 - s)), noise, full(sigma))
        return x
