"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's host-side sampling math: beta schedules, the spaced DDPM sampler,
DPM-Solver++(2M) multistep, and the mixture-of-diffusers tiling.  Model evaluations are delegated to a
``model(x, t, cond) -> eps|v`` callable, exactly like the reference samplers do.
"""
import math
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

T = torch.Tensor


# ---------------------------------------------------------------- schedules
def make_betas(linear_start=1e-4, linear_end=2e-2, timesteps=1000, zero_snr=False, **_) -> np.ndarray:
    """reference gaussian_diffusion.py:9-36 ("linear" schedule) and 49-72 (zero terminal SNR)."""
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
    if zero_snr:
        b = torch.from_numpy(betas)
        abar_sqrt = (1 - b).cumprod(0).sqrt()
        a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
        abar_sqrt = (abar_sqrt - aT) * (a0 / (a0 - aT))
        abar = abar_sqrt ** 2
        alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
        betas = (1 - alphas).numpy()
    return betas


def q_sample(betas: np.ndarray, x0: T, t: T, noise: T) -> T:
    """reference gaussian_diffusion.py:117-129 (fp32 buffers)."""
    ac = np.cumprod(1.0 - betas, axis=0)
    a = torch.tensor(np.sqrt(ac), dtype=torch.float32)[t].view(-1, 1, 1, 1)
    s = torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32)[t].view(-1, 1, 1, 1)
    return a * x0 + s * noise


def space_timesteps(num_timesteps: int, count: int) -> List[int]:
    """reference spaced_sampler.py:14-64 for a single section "<count>" (Python banker's round)."""
    if count <= 1:
        stride = 1.0
    else:
        stride = (num_timesteps - 1) / (count - 1)
    if num_timesteps < count:
        raise ValueError(f"cannot divide section of {num_timesteps} steps into {count}")
    cur, out = 0.0, []
    for _ in range(count):
        out.append(round(cur))
        cur += stride
    return sorted(set(out))


def spaced_tables(betas: np.ndarray, steps: int) -> Dict[str, np.ndarray]:
    """reference spaced_sampler.py:77-116. Returns float64 tables + the int timestep list."""
    ac_train = np.cumprod(1.0 - betas, axis=0)
    used = space_timesteps(len(betas), steps)
    nb, last = [], 1.0
    for i in used:
        nb.append(1 - ac_train[i] / last)
        last = ac_train[i]
    nb = np.array(nb, dtype=np.float64)
    alphas = 1.0 - nb
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    with np.errstate(divide="ignore", invalid="ignore"):
        tb = dict(
            timesteps=np.array(used, dtype=np.int32),
            sqrt_alphas_cumprod=np.sqrt(ac),
            sqrt_one_minus_alphas_cumprod=np.sqrt(1 - ac),
            sqrt_recip_alphas_cumprod=np.sqrt(1.0 / ac),
            sqrt_recipm1_alphas_cumprod=np.sqrt(1.0 / ac - 1),
            posterior_variance=nb * (1.0 - ac_prev) / (1.0 - ac),
            posterior_mean_coef1=nb * np.sqrt(ac_prev) / (1.0 - ac),
            posterior_mean_coef2=(1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
        )
    return tb


def cfg_scale_at(rescale_cfg: bool, cfg_scale: float, model_t: int) -> float:
    """reference sampler.py:31-38."""
    if rescale_cfg and cfg_scale > 1:
        return 1 + cfg_scale * ((1 - math.cos(math.pi * ((1000 - model_t) / 1000) ** 5.0)) / 2)
    return cfg_scale


# ---------------------------------------------------------------- tiling (mixture of diffusers)
def sliding_windows(h: int, w: int, size: int, stride: int) -> List[Tuple[int, int, int, int]]:
    """reference utils/common.py:123-138 (edge-flush extra window)."""
    his = list(range(0, h - size + 1, stride))
    if (h - size) % stride != 0:
        his.append(h - size)
    wis = list(range(0, w - size + 1, stride))
    if (w - size) % stride != 0:
        wis.append(w - size)
    return [(hi, hi + size, wi, wi + size) for hi in his for wi in wis]


def gaussian_weights(tile_w: int, tile_h: int) -> np.ndarray:
    """reference utils/common.py:142-169 — note x midpoint (w-1)/2 but y midpoint h/2."""
    var = 0.01
    mx = (tile_w - 1) / 2
    xp = [np.exp(-(x - mx) * (x - mx) / (tile_w * tile_w) / (2 * var)) / np.sqrt(2 * np.pi * var)
          for x in range(tile_w)]
    my = tile_h / 2
    yp = [np.exp(-(y - my) * (y - my) / (tile_h * tile_h) / (2 * var)) / np.sqrt(2 * np.pi * var)
          for y in range(tile_h)]
    return np.outer(yp, xp)


def make_tiled_fn(fn: Callable, size: int, stride: int, scale: int = 1, gaussian: bool = True) -> Callable:
    """reference utils/common.py:172-232 (scale_type="up")."""

    def tiled(x: T, *args, **kwargs) -> T:
        b, c, h, w = x.shape
        out = torch.zeros((b, c, h * scale, w * scale), dtype=x.dtype)
        count = torch.zeros_like(out, dtype=torch.float32)
        ws = size * scale
        wt = gaussian_weights(ws, ws)[None, None] if gaussian else np.ones((1, 1, ws, ws))
        wt = torch.tensor(wt, dtype=x.dtype)
        for hi, he, wi, we in sliding_windows(h, w, size, stride):
            if len(args) or len(kwargs):
                kwargs.update(dict(hi=hi, hi_end=he, wi=wi, wi_end=we))
            out[..., hi * scale:he * scale, wi * scale:we * scale] += fn(x[..., hi:he, wi:we], *args, **kwargs) * wt
            count[..., hi * scale:he * scale, wi * scale:we * scale] += wt
        return out / count

    return tiled


def tile_model(model: Callable, size: int, stride: int) -> Callable:
    """reference spaced_sampler.py:204-219 / dpms_sampler.py:56-71: crop c_img with the tile."""
    return make_tiled_fn(
        lambda xt, t, cond, hi, hi_end, wi, wi_end: model(
            xt, t, {"c_txt": cond["c_txt"], "c_img": cond["c_img"][..., hi:hi_end, wi:wi_end]}),
        size, stride)


# ---------------------------------------------------------------- spaced DDPM sampler
def spaced_sample(model: Callable, betas: np.ndarray, parameterization: str, steps: int, x_T: T,
                  cond: dict, uncond: Optional[dict], cfg_scale: float, rescale_cfg: bool = False,
                  noise_fn: Callable = torch.randn_like, tiled=False, tile_size=-1, tile_stride=-1) -> T:
    """reference spaced_sampler.py:144-245."""
    tb = spaced_tables(betas, steps)
    f32 = {k: torch.tensor(v, dtype=torch.float32) for k, v in tb.items() if k != "timesteps"}
    if tiled:
        model = tile_model(model, tile_size, tile_stride)
    x = x_T
    bs = x.shape[0]
    total = len(tb["timesteps"])
    for i, step in enumerate(np.flip(tb["timesteps"])):
        model_t = torch.full((bs,), int(step), dtype=torch.long)
        ti = total - i - 1
        s = cfg_scale_at(rescale_cfg, cfg_scale, int(step))
        if uncond is None or s == 1.0:
            out = model(x, model_t, cond)
        else:
            oc = model(x, model_t, cond)
            ou = model(x, model_t, uncond)
            out = ou + s * (oc - ou)
        if parameterization == "eps":
            x0 = f32["sqrt_recip_alphas_cumprod"][ti] * x - f32["sqrt_recipm1_alphas_cumprod"][ti] * out
        else:
            x0 = f32["sqrt_alphas_cumprod"][ti] * x - f32["sqrt_one_minus_alphas_cumprod"][ti] * out
        mean = f32["posterior_mean_coef1"][ti] * x0 + f32["posterior_mean_coef2"][ti] * x
        noise = noise_fn(x)
        x = mean + (1.0 if ti != 0 else 0.0) * torch.sqrt(f32["posterior_variance"][ti]) * noise
    return x


# ---------------------------------------------------------------- DPM-Solver++ (2M), discrete VP schedule
class VPSchedule:
    """reference dpm_solver_pytorch.py:99-168 (schedule='discrete'), interpolate_fn 1273-1312."""

    def __init__(self, betas: np.ndarray):
        b = torch.tensor(betas, dtype=torch.float32)  # DPMSolverSampler registers betas as fp32 (dpms_sampler.py:40)
        log_alphas = 0.5 * torch.log(1 - b).cumsum(dim=0)
        log_sigmas = 0.5 * torch.log(1.0 - torch.exp(2.0 * log_alphas))
        lambs = log_alphas - log_sigmas
        idx = int(torch.searchsorted(torch.flip(lambs, [0]), torch.tensor(-5.1)))
        if idx > 0:
            log_alphas = log_alphas[:-idx]
        self.log_alpha = log_alphas.to(torch.float32)
        self.total_N = self.log_alpha.shape[0]
        self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1)[1:].to(torch.float32)

    def log_mean_coeff(self, t: T) -> T:
        """piecewise-linear interpolation of log(alpha) at t, linear extrapolation outside the keypoints."""
        xp, yp = self.t_array, self.log_alpha
        K = xp.shape[0]
        t = t.reshape(-1)
        idx = torch.searchsorted(xp, t, right=False)  # number of keypoints strictly below t
        # segment start index following the reference's sort-based rule
        start = torch.where(idx == 0, torch.zeros_like(idx), torch.where(idx == K, torch.full_like(idx, K - 2), idx - 1))
        x0, x1 = xp[start], xp[start + 1]
        y0, y1 = yp[start], yp[start + 1]
        return y0 + (t - x0) * (y1 - y0) / (x1 - x0)

    def alpha(self, t):
        return torch.exp(self.log_mean_coeff(t))

    def std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.log_mean_coeff(t)))

    def lam(self, t):
        lm = self.log_mean_coeff(t)
        return lm - 0.5 * torch.log(1.0 - torch.exp(2.0 * lm))


def dpmpp_2m_sample(model: Callable, betas: np.ndarray, parameterization: str, steps: int, x_T: T,
                    cond: dict, uncond: Optional[dict], cfg_scale: float, order: int = 2,
                    tiled=False, tile_size=-1, tile_stride=-1) -> T:
    """reference dpms_sampler.py:42-101 with model_spec "dpm++_m{1,2}": model_wrapper (batched CFG,
    dpm_solver_pytorch.py:273-349), data_prediction_fn 451-459, first update 565-610, multistep second update
    814-870, sample (multistep branch) 1189-1233."""
    ns = VPSchedule(betas)
    if tiled:
        model = tile_model(model, tile_size, tile_stride)
    bs = x_T.shape[0]

    def x0_pred(x: T, t: T) -> T:
        tc = t.expand(bs)
        t_in = (tc - 1.0 / ns.total_N) * 1000.0

        def raw(xx, tt_c, tt_in, c):
            out = model(xx, tt_in, c)
            if parameterization == "v":
                a, s = ns.alpha(tt_c).view(-1, 1, 1, 1), ns.std(tt_c).view(-1, 1, 1, 1)
                return a * out + s * xx
            return out

        if uncond is None or cfg_scale == 1.0:
            eps = raw(x, tc, t_in, cond)
        else:
            c_in = {k: torch.cat([uncond[k], cond[k]]) for k in cond}
            e_u, e_c = raw(torch.cat([x] * 2), torch.cat([tc] * 2), torch.cat([t_in] * 2), c_in).chunk(2)
            eps = e_u + cfg_scale * (e_c - e_u)
        a, s = ns.alpha(t), ns.std(t)
        return (x - s * eps) / a

    ts = torch.linspace(1.0, 1.0 / ns.total_N, steps + 1)
    t_prev = [ts[0].reshape(1)]
    m_prev = [x0_pred(x_T, t_prev[0])]
    x = x_T

    def first(x, s, t, ms):
        h = ns.lam(t) - ns.lam(s)
        return ns.std(t) / ns.std(s) * x - ns.alpha(t) * torch.expm1(-h) * ms

    def second(x, t):
        l1, l0, lt = ns.lam(t_prev[-2]), ns.lam(t_prev[-1]), ns.lam(t)
        h0, h = l0 - l1, lt - l0
        r0 = h0 / h
        D1 = (1.0 / r0) * (m_prev[-1] - m_prev[-2])
        phi = torch.expm1(-h)
        a_t = ns.alpha(t)
        return (ns.std(t) / ns.std(t_prev[-1])) * x - (a_t * phi) * m_prev[-1] - 0.5 * (a_t * phi) * D1

    for step in range(1, order):
        t = ts[step].reshape(1)
        x = first(x, t_prev[-1], t, m_prev[-1])
        t_prev.append(t)
        m_prev.append(x0_pred(x, t))
    for step in range(order, steps + 1):
        t = ts[step].reshape(1)
        so = min(order, steps + 1 - step) if steps < 10 else order
        x = first(x, t_prev[-1], t, m_prev[-1]) if so == 1 else second(x, t)
        for i in range(order - 1):
            t_prev[i], m_prev[i] = t_prev[i + 1], m_prev[i + 1]
        t_prev[-1] = t
        if step < steps:
            m_prev[-1] = x0_pred(x, t)
    return x
