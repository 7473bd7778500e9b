"""DPM-Solver++ multistep sampler (reference sampler/dpms_sampler.py:17-101 driving the vendored
dpm_solver_pytorch.py: NoiseScheduleVP 99-168, model_wrapper 273-349, data_prediction_fn 451-459, first update
565-594, multistep second update 814-849, sample/multistep 1189-1233).

Only the path the DiffBIR configs use is implemented: `dpm++_m1` / `dpm++_m2`, discrete VP schedule,
`time_uniform` steps, classifier-free guidance batched as [uncond, cond].  Scalar schedule quantities are
computed on the host in float32 torch (what the reference does on CPU); tensor updates are fused f32
`dbir_lincomb4` launches.
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import ops
from .sampler import Sampler


class NoiseScheduleVP:
    """Discrete VP schedule with piecewise-linear log(alpha_t) (dpm_solver_pytorch.py:99-168, 1273-1312)."""

    def __init__(self, betas: torch.Tensor):
        la = 0.5 * torch.log(1 - betas).cumsum(dim=0)
        lam = la - 0.5 * torch.log(1.0 - torch.exp(2.0 * la))
        drop = int(torch.searchsorted(torch.flip(lam, [0]), torch.tensor(-5.1)))  # numerical_clip_alpha
        if drop > 0:
            la = la[:-drop]
        self.log_alpha = la.to(torch.float32)
        self.total_N = self.log_alpha.shape[0]
        self.T = 1.0
        self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1)[1:].to(torch.float32)

    def marginal_log_mean_coeff(self, t: torch.Tensor) -> torch.Tensor:
        xp, yp, K = self.t_array, self.log_alpha, self.total_N
        t = t.reshape(-1)
        below = torch.searchsorted(xp, t, right=False)
        lo = torch.where(below == 0, torch.zeros_like(below),
                         torch.where(below == K, torch.full_like(below, K - 2), below - 1))
        x0, x1, y0, y1 = xp[lo], xp[lo + 1], yp[lo], yp[lo + 1]
        return y0 + (t - x0) * (y1 - y0) / (x1 - x0)

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        lm = self.marginal_log_mean_coeff(t)
        return lm - 0.5 * torch.log(1.0 - torch.exp(2.0 * lm))


class DPMSolverSampler(Sampler):
    def __init__(self, betas: np.ndarray, parameterization: str, rescale_cfg: bool, model_spec: str):
        super().__init__(betas, parameterization, rescale_cfg)
        if parameterization not in ("eps", "v"):
            raise ValueError(parameterization)
        solver, (method, order) = model_spec.split("_")
        self.solver_type = {"dpm": "dpmsolver", "dpm++": "dpmsolver++"}[solver]
        self.method = {"s": "singlestep", "m": "multistep"}[method]
        self.order = {"1": 1, "2": 2, "3": 3}[order]
        if self.solver_type != "dpmsolver++" or self.method != "multistep" or self.order > 2:
            raise NotImplementedError(f"{model_spec}: the engine implements dpm++_m1 / dpm++_m2 (DiffBIR's configs); "
                                      "singlestep / 3rd-order / dpmsolver variants are out of scope")
        self.betas = torch.tensor(betas, dtype=torch.float32)
        if rescale_cfg:
            raise NotImplementedError("rescale_cfg with DPM-Solver only works at batch 1 in the reference "
                                      "(dpm_solver_pytorch.py:343-348, math.cos on a tensor); unsupported")

    @torch.no_grad()
    def sample(self, model, device: str, steps: int, x_size: Tuple[int], cond: Dict[str, torch.Tensor],
               uncond: Optional[Dict[str, torch.Tensor]], cfg_scale: float, tiled: bool = False, tile_size: int = -1,
               tile_stride: int = -1, x_T: Optional[torch.Tensor] = None, progress: bool = True) -> torch.Tensor:
        ns = NoiseScheduleVP(self.betas)
        fwd = self._tiled(model.forward, tile_size, tile_stride) if tiled else model.forward
        bs = x_size[0]
        if x_T is None:
            x_T = self._randn(x_size, device)
        x = x_T.to(device=device, dtype=torch.float32).contiguous()
        use_cfg = not (uncond is None or cfg_scale == 1.0)
        if use_cfg:
            cond2 = self._cfg_batch(cond, uncond, bs)
        g = float(cfg_scale)
        full = lambda v: torch.full((bs,), float(v), device=device, dtype=torch.float32)

        def x0_pred(xc: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
            """data prediction with CFG: x0 = (x - sigma*eps)/alpha, eps from eps- or v-output."""
            t_in = float((t - 1.0 / ns.total_N) * 1000.0)
            a, s = float(ns.marginal_alpha(t)), float(ns.marginal_std(t))
            if use_cfg:
                tt = torch.full((2 * bs,), t_in, device=device, dtype=torch.float32)
                o = fwd(torch.cat([xc, xc], dim=0), tt, cond2)
                ou, oc = o[:bs].contiguous(), o[bs:].contiguous()
            else:
                oc, ou = fwd(xc, torch.full((bs,), t_in, device=device, dtype=torch.float32), cond).contiguous(), None
            if self.parameterization == "v":   # eps = a*out + s*x  ->  x0 = x*(1-s^2)/a - s*out
                cx, co = (1.0 - s * s) / a, -s
            else:                               # x0 = x/a - (s/a)*out
                cx, co = 1.0 / a, -s / a
            if ou is None:
                return ops.lincomb4(xc, full(cx), oc, full(co))
            return ops.lincomb4(xc, full(cx), ou, full(co * (1.0 - g)), oc, full(co * g))

        ts = torch.linspace(ns.T, 1.0 / ns.total_N, steps + 1)
        order = self.order
        assert steps >= order
        t_prev = [ts[0].reshape(1)]
        m_prev = [x0_pred(x, t_prev[0])]

        def first(xc, s_, t_, ms):
            h = ns.marginal_lambda(t_) - ns.marginal_lambda(s_)
            cx = float(ns.marginal_std(t_) / ns.marginal_std(s_))
            cm = float(-ns.marginal_alpha(t_) * torch.expm1(-h))
            return ops.lincomb4(xc, full(cx), ms, full(cm))

        def second(xc, t_):
            l1, l0, lt = (ns.marginal_lambda(v) for v in (t_prev[-2], t_prev[-1], t_))
            h0, h = l0 - l1, lt - l0
            r0 = h0 / h
            ap = ns.marginal_alpha(t_) * torch.expm1(-h)
            cx = float(ns.marginal_std(t_) / ns.marginal_std(t_prev[-1]))
            c0 = float(-ap - 0.5 * ap / r0)
            c1 = float(0.5 * ap / r0)
            return ops.lincomb4(xc, full(cx), m_prev[-1], full(c0), m_prev[-2], full(c1))

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
