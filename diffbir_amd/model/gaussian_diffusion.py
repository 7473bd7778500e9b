"""Diffusion schedule (reference gaussian_diffusion.py:75-129) — host-side float64 numpy, identical arithmetic."""
import numpy as np
import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule != "linear":
        raise NotImplementedError(f"beta schedule {schedule!r}: only 'linear' is on the DiffBIR inference path")
    return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2


def enforce_zero_terminal_snr(betas: np.ndarray) -> np.ndarray:
    """arXiv:2305.08891 rescale (reference gaussian_diffusion.py:49-72), computed in torch float64 like the reference."""
    b = torch.from_numpy(betas)
    s = (1 - b).cumprod(0).sqrt()
    s0, sT = s[0].clone(), s[-1].clone()
    s = (s - sT) * (s0 / (s0 - sT))
    abar = s ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return (1 - alphas).numpy()


class Diffusion:
    def __init__(self, timesteps=1000, beta_schedule="linear", loss_type="l2", linear_start=1e-4, linear_end=2e-2,
                 cosine_s=8e-3, parameterization="eps", zero_snr=False):
        assert parameterization in ("eps", "x0", "v")
        self.num_timesteps = timesteps
        self.parameterization = parameterization
        self.zero_snr = zero_snr
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start, linear_end, cosine_s)
        if zero_snr:
            betas = enforce_zero_terminal_snr(betas)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.betas = betas
        self.sqrt_alphas_cumprod = torch.tensor(np.sqrt(ac), dtype=torch.float32)
        self.sqrt_one_minus_alphas_cumprod = torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32)

    def to(self, device):
        self.sqrt_alphas_cumprod = self.sqrt_alphas_cumprod.to(device)
        self.sqrt_one_minus_alphas_cumprod = self.sqrt_one_minus_alphas_cumprod.to(device)
        return self

    def q_sample(self, x_start, t, noise):
        sh = (-1,) + (1,) * (x_start.dim() - 1)
        a = self.sqrt_alphas_cumprod.to(x_start.device)[t].view(sh)
        s = self.sqrt_one_minus_alphas_cumprod.to(x_start.device)[t].view(sh)
        return a * x_start + s * noise
