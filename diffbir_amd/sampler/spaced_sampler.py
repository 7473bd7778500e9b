"""Spaced DDPM sampler (reference sampler/spaced_sampler.py:14-245) on the engine.

Host-side schedule math is float64 numpy exactly as in the reference (tables registered as f32).  Per step the
engine runs ONE batched network evaluation for classifier-free guidance (uncond || cond at batch 2B — the
reference does two batch-B evaluations, identical per sample) and ONE fused f32 kernel for
CFG mix + x0 prediction + posterior mean + noise (`dbir_spaced_step`).
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import ops
from .sampler import Sampler


def space_timesteps(num_timesteps: int, section_counts) -> set:
    """reference spaced_sampler.py:14-64 (incl. the "ddimN" form and Python's round-half-even)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += frac
        start += size
    return set(steps)


class SpacedSampler(Sampler):
    def make_schedule(self, num_steps: int) -> None:
        """reference spaced_sampler.py:77-116."""
        used = space_timesteps(self.num_timesteps, str(num_steps))
        betas, last = [], 1.0
        for i, ac in enumerate(self.training_alphas_cumprod):
            if i in used:
                betas.append(1 - ac / last)
                last = ac
        self.timesteps = np.array(sorted(used), dtype=np.int32)
        betas = np.array(betas, dtype=np.float64)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        with np.errstate(divide="ignore", invalid="ignore"):  # zero terminal SNR: 1/0 at the last entry (unused by v)
            tb = dict(
                sqrt_alphas_cumprod=np.sqrt(ac), sqrt_one_minus_alphas_cumprod=np.sqrt(1 - ac),
                sqrt_recip_alphas_cumprod=np.sqrt(1.0 / ac), sqrt_recipm1_alphas_cumprod=np.sqrt(1.0 / ac - 1),
                posterior_variance=betas * (1.0 - ac_prev) / (1.0 - ac),
                posterior_mean_coef1=betas * np.sqrt(ac_prev) / (1.0 - ac),
                posterior_mean_coef2=(1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))
        self.tables = {k: torch.tensor(v, dtype=torch.float32) for k, v in tb.items()}

    @torch.no_grad()
    def sample(self, model, device: str, steps: int, x_size: Tuple[int], cond: Dict[str, torch.Tensor],
               uncond: Optional[Dict[str, torch.Tensor]], cfg_scale: float, tiled: bool = False, tile_size: int = -1,
               tile_stride: int = -1, x_T: Optional[torch.Tensor] = None, progress: bool = True) -> torch.Tensor:
        self.make_schedule(steps)
        bs = x_size[0]
        fwd = model.forward
        if tiled:
            fwd = self._tiled(model.forward, tile_size, tile_stride)
        if x_T is None:
            x_T = self._randn(x_size, device)
        x = x_T.to(device=device, dtype=torch.float32).contiguous()
        tb = {k: v.to(device) for k, v in self.tables.items()}
        if self.parameterization == "eps":
            k_x, k_o = tb["sqrt_recip_alphas_cumprod"], tb["sqrt_recipm1_alphas_cumprod"]
        else:
            k_x, k_o = tb["sqrt_alphas_cumprod"], tb["sqrt_one_minus_alphas_cumprod"]
        sd = torch.sqrt(tb["posterior_variance"])
        sd[0] = 0.0  # nonzero_mask (t != 0)
        per_b = lambda v: v[:, None].expand(-1, bs).contiguous()
        k_x, k_o, c1, c2, sd = (per_b(v) for v in (k_x, k_o, tb["posterior_mean_coef1"], tb["posterior_mean_coef2"], sd))
        use_cfg = not (uncond is None or cfg_scale == 1.0)
        if use_cfg:
            cond2 = self._cfg_batch(cond, uncond, bs)
        # the per-step host timestep (an engine extension read by ControlLDM.forward) goes into PRIVATE copies: the caller's
        # dict is never written, so a cond reused by another sampler / a direct model call cannot carry a stale timestep
        cond = dict(cond)
        if use_cfg:
            cond2 = dict(cond2)
        total = len(self.timesteps)
        it = np.flip(self.timesteps)
        if progress:
            from tqdm import tqdm
            it = tqdm(it, total=total)
        for i, step in enumerate(it):
            ti = total - i - 1
            model_t = torch.full((bs,), int(step), device=device, dtype=torch.float32)
            s = self.get_cfg_scale(cfg_scale, int(step))
            cond["t_host"] = float(int(step))  # engine extension: every element of model_t is this host scalar
            if use_cfg:
                cond2["t_host"] = cond["t_host"]
            if use_cfg and s != 1.0:
                o = fwd(torch.cat([x, x], dim=0), torch.cat([model_t, model_t]), cond2)
                ou, oc = o[:bs], o[bs:]
            else:
                oc, ou = fwd(x, model_t, cond), None
            noise = self._randn(x.shape, device)
            x = ops.spaced_step(x, oc.contiguous(), None if ou is None else ou.contiguous(), noise, float(s),
                                k_x[ti], k_o[ti], c1[ti], c2[ti], sd[ti])
        return x
