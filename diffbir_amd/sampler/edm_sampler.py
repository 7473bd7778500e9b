"""EDM / k-diffusion samplers (reference sampler/edm_sampler.py:26-188 driving sampler/k_diffusion.py:123-707) on the
engine: `edm_euler`, `edm_euler_a`, `edm_heun`, `edm_dpm_2`, `edm_dpm_2_a`, `edm_lms`, `edm_dpm++_2s_a`,
`edm_dpm++_sde`, `edm_dpm++_2m`, `edm_dpm++_2m_sde`, `edm_dpm++_3m_sde` (the reference CLI's default).

The algorithms are the published ones (Karras et al. 2022, Alg. 2; Lu et al. 2022, DPM-Solver++; Crowson's k-diffusion
formulation), restated over three primitives:
  * `denoise(x, sigma)` — the VP preconditioning of reference edm_sampler.py:104-141: c_in / c_out / c_skip from sigma, the
    model timestep = the training timestep whose table sigma is nearest (f32 arithmetic like the reference), ONE batched
    network evaluation (uncond || cond at batch 2B) and one fused f32 kernel for  c_skip*x + c_out*(u + s*(c - u));
  * `axpy`-style updates on the f32 latent (`dbir_lincomb4`);
  * a noise source.  Deterministic / ancestral solvers draw `randn_like(x)` from the device generator in exactly the
    reference's order (also the unused `eps` of the churn-free Euler / Heun / DPM-2 steps).  The SDE solvers need a
    Brownian motion W over sigma time: the reference uses torchsde.BrownianTree through k-diffusion's
    BrownianTreeNoiseSampler (k_diffusion.py:70-119); the engine's default is `brownian.BrownianTreeNoise`, a restatement
    of that tree (seed drawn from torch's global CPU generator like the reference's, numpy SeedSequence keyed by tree
    position, per-node torch generators) — unverified against a torchsde install, see sampler/brownian.py.
    `sampler.brownian` may be replaced by a factory (x, randn, sigma_min, sigma_max) -> callable(sigma, sigma_next):
    `BrownianPath` below (a lazily refined bridge on the sampler's own noise source) or a test stand-in.
All scalar schedule math is done on the host in float32, mirroring the reference's f32 buffers.
"""
import math
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import ops
from .sampler import Sampler

f32 = np.float32


class BrownianPath:
    """W(t) on [t_lo, t_hi], sampled lazily: a new time between two known ones is drawn from the Brownian bridge between
    them, outside the known range from an independent increment.  Returns (W(t1) - W(t0)) / sqrt(|t1 - t0|), the
    convention of k-diffusion's BrownianTreeNoiseSampler (k_diffusion.py:97-119)."""

    def __init__(self, x: torch.Tensor, randn: Callable, *_sigma_range):
        self.shape, self.randn, self.like = tuple(x.shape), randn, x
        self.t: List[float] = []
        self.w: List[torch.Tensor] = []

    def _at(self, t: float) -> torch.Tensor:
        import bisect
        if not self.t:
            self.t.append(t)
            self.w.append(torch.zeros_like(self.like, dtype=torch.float32))
            return self.w[0]
        i = bisect.bisect_left(self.t, t)
        if i < len(self.t) and self.t[i] == t:
            return self.w[i]
        if i == 0:
            w = self.w[0] - self.randn(self.shape) * math.sqrt(self.t[0] - t)
        elif i == len(self.t):
            w = self.w[-1] + self.randn(self.shape) * math.sqrt(t - self.t[-1])
        else:
            ta, tb = self.t[i - 1], self.t[i]
            lam = (t - ta) / (tb - ta)
            w = self.w[i - 1] * (1 - lam) + self.w[i] * lam + self.randn(self.shape) * math.sqrt((t - ta) * (tb - t) / (tb - ta))
        self.t.insert(i, t)
        self.w.insert(i, w)
        return w

    def __call__(self, sigma: float, sigma_next: float) -> torch.Tensor:
        t0, t1 = float(sigma), float(sigma_next)
        return (self._at(t1) - self._at(t0)) / math.sqrt(abs(t1 - t0))


def _ancestral(s_from, s_to, eta: float):
    """sigma_down, sigma_up of an ancestral step (k_diffusion.py:56-63) in float32, operation for operation: at the first
    step (sigma_from = 1e4) `sigma_from^2 - sigma_to^2` rounds to sigma_from^2, sigma_up = sigma_to and sigma_down is
    exactly 0, which selects the solvers' Euler branch (and no noise) in the reference — a float64 evaluation would not."""
    s_from, s_to = f32(s_from), f32(s_to)
    if not eta:
        return s_to, f32(0.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        up = min(s_to, f32(eta) * (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** f32(0.5))
        down = (s_to ** 2 - up ** 2) ** f32(0.5)
    return f32(down), f32(up)


def _log(v):
    with np.errstate(divide="ignore"):
        return np.log(f32(v))


class EDMSampler(Sampler):
    SOLVERS = ("euler", "euler_a", "heun", "dpm_2", "dpm_2_a", "lms", "dpm++_2s_a", "dpm++_sde", "dpm++_2m",
               "dpm++_2m_sde", "dpm++_3m_sde")

    def __init__(self, betas: np.ndarray, parameterization: str, rescale_cfg: bool, solver_type: str, s_churn: float,
                 s_tmin: float, s_tmax: float, s_noise: float, eta: float, order: int):
        super().__init__(betas, parameterization, rescale_cfg)
        name = solver_type[len("edm_"):]
        if name not in self.SOLVERS:
            raise KeyError(name)
        self.solver = name
        self.hp = dict(s_churn=s_churn, s_tmin=s_tmin, s_tmax=s_tmax, s_noise=s_noise, eta=eta, order=order)
        # engine extension: factory (x, randn, sigma_min, sigma_max) -> callable(sigma, sigma_next) for the SDE solvers;
        # None = the restated torchsde tree (brownian.BrownianTreeNoise, what the reference builds at k_diffusion.py:551,623,665)
        self.brownian: Optional[Callable] = None

    # ---------------------------------------------------------------- schedule (reference edm_sampler.py:86-98)
    def make_schedule(self, steps: int) -> None:
        ts = np.linspace(len(self.training_alphas_cumprod) - 1, 0, steps, endpoint=False).astype(int)
        ac = self.training_alphas_cumprod[ts].copy()
        ac[0] = 1e-8   # avoid the divide-by-zero of zero-terminal-SNR schedules
        self.sigmas = np.append(((1 - ac) / ac) ** 0.5, 0).astype(np.float32)
        self.timesteps = np.append(ts, 0).astype(np.int64)

    # ---------------------------------------------------------------- the denoiser
    def _denoiser(self, fwd, cond, uncond, cfg_scale: float, bs: int, device):
        use_cfg = not (uncond is None or cfg_scale == 1.0)
        cond2 = self._cfg_batch(cond, uncond, bs) if use_cfg else None
        full = lambda v: torch.full((bs,), float(v), device=device, dtype=torch.float32)

        def denoise(x: torch.Tensor, sigma) -> torch.Tensor:
            sg = f32(sigma)
            if self.parameterization == "eps":
                c_skip, c_out = f32(1.0), -sg
            else:
                c_skip, c_out = f32(1.0) / (sg * sg + f32(1.0)), -sg / (sg * sg + f32(1.0)) ** f32(0.5)
            c_in = f32(1.0) / (sg * sg + f32(1.0)) ** f32(0.5)
            t_idx = int(np.abs(sg - self.sigmas).argmin())          # nearest table sigma, f32 like the reference
            step = int(self.timesteps[t_idx])
            s = float(self.get_cfg_scale(cfg_scale, step))
            model_t = torch.full((bs,), step, device=device, dtype=torch.float32)
            xin = ops.lincomb4(x, full(c_in))
            if use_cfg:
                o = fwd(torch.cat([xin, xin], dim=0), torch.cat([model_t, model_t]), cond2)
                return ops.lincomb4(x, full(c_skip), o[bs:].contiguous(), full(c_out * s), o[:bs].contiguous(),
                                    full(c_out * (1.0 - s)))
            return ops.lincomb4(x, full(c_skip), fwd(xin, model_t, cond).contiguous(), full(c_out))

        return denoise

    # ---------------------------------------------------------------- sampling
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
        x_T = x_T.to(device=device, dtype=torch.float32).contiguous()
        sig = [f32(s) for s in self.sigmas]   # float32 scalars: host math mirrors the reference's f32 tensors
        self._full = lambda v: torch.full((bs,), float(v), device=device, dtype=torch.float32)
        self._noise = lambda: self._randn(x_T.shape, device)
        x = ops.lincomb4(x_T, self._full(math.sqrt(1.0 + sig[0] ** 2)))
        den = self._denoiser(fwd, cond, uncond, cfg_scale, bs, device)
        return getattr(self, "_solve_" + self.solver.replace("++", "pp"))(den, x, sig)

    # small helpers: out = a*x + b*y (+ c*z)
    def _lin(self, x, a, y=None, b=0.0, z=None, c=0.0):
        F = self._full
        if y is None:
            return ops.lincomb4(x, F(a))
        if z is None:
            return ops.lincomb4(x, F(a), y, F(b))
        return ops.lincomb4(x, F(a), y, F(b), z, F(c))

    def _euler_to(self, x, den_x, s_from, s_to):
        """x + (x - denoised) / s_from * (s_to - s_from)"""
        r = (s_to - s_from) / s_from
        return self._lin(x, 1.0 + r, den_x, -r)

    def _churn(self, x, s, n):
        """Karras Alg. 2 stochastic churn; one Gaussian draw per step whether it is used or not."""
        hp = self.hp
        gamma = min(hp["s_churn"] / n, 2 ** 0.5 - 1) if hp["s_tmin"] <= s <= hp["s_tmax"] else 0.0
        eps = self._noise()
        s_hat = f32(s * f32(gamma + 1))
        if gamma > 0:
            x = self._lin(x, 1.0, eps, hp["s_noise"] * float((s_hat ** 2 - s ** 2) ** f32(0.5)))
        return x, s_hat

    # ---- Karras et al. Alg. 2, first order (k_diffusion.py:123-140)
    def _solve_euler(self, den, x, sig):
        n = len(sig) - 1
        for i in range(n):
            x, s_hat = self._churn(x, sig[i], n)
            x = self._euler_to(x, den(x, s_hat), s_hat, sig[i + 1])
        return x

    # ---- ancestral Euler (k_diffusion.py:143-160)
    def _solve_euler_a(self, den, x, sig):
        for i in range(len(sig) - 1):
            d = den(x, sig[i])
            down, up = _ancestral(sig[i], sig[i + 1], self.hp["eta"])
            x = self._euler_to(x, d, sig[i], down)
            if sig[i + 1] > 0:
                x = self._lin(x, 1.0, self._noise(), self.hp["s_noise"] * up)
        return x

    # ---- Karras Alg. 2, Heun (k_diffusion.py:163-189)
    def _solve_heun(self, den, x, sig):
        n = len(sig) - 1
        for i in range(n):
            x, s_hat = self._churn(x, sig[i], n)
            d0 = den(x, s_hat)
            dt = sig[i + 1] - s_hat
            if sig[i + 1] == 0:
                x = self._euler_to(x, d0, s_hat, sig[i + 1])
            else:
                x2 = self._euler_to(x, d0, s_hat, sig[i + 1])
                d2 = den(x2, sig[i + 1])
                # x + dt/2 * ((x - d0)/s_hat + (x2 - d2)/s_next)
                a, b = 0.5 * dt / s_hat, 0.5 * dt / sig[i + 1]
                x = ops.lincomb4(x, self._full(1.0 + a), d0, self._full(-a), x2, self._full(b), d2, self._full(-b))
        return x

    # ---- DPM-Solver-2 flavoured step of k-diffusion (k_diffusion.py:192-220)
    def _solve_dpm_2(self, den, x, sig):
        n = len(sig) - 1
        for i in range(n):
            x, s_hat = self._churn(x, sig[i], n)
            d0 = den(x, s_hat)
            if sig[i + 1] == 0:
                x = self._euler_to(x, d0, s_hat, sig[i + 1])
            else:
                s_mid = np.exp(_log(s_hat) + f32(0.5) * (_log(sig[i + 1]) - _log(s_hat)))   # log().lerp(.., 0.5).exp()
                x2 = self._euler_to(x, d0, s_hat, s_mid)
                d2 = den(x2, s_mid)
                r = (sig[i + 1] - s_hat) / s_mid            # x + (x2 - d2)/s_mid * dt_2
                x = ops.lincomb4(x, self._full(1.0), x2, self._full(r), d2, self._full(-r))
        return x

    # ---- ancestral DPM-Solver-2 (k_diffusion.py:223-249)
    def _solve_dpm_2_a(self, den, x, sig):
        for i in range(len(sig) - 1):
            d0 = den(x, sig[i])
            down, up = _ancestral(sig[i], sig[i + 1], self.hp["eta"])
            if down == 0:
                x = self._euler_to(x, d0, sig[i], down)
            else:
                s_mid = np.exp(_log(sig[i]) + f32(0.5) * (_log(down) - _log(sig[i])))
                x2 = self._euler_to(x, d0, sig[i], s_mid)
                d2 = den(x2, s_mid)
                r = (down - sig[i]) / s_mid
                x = ops.lincomb4(x, self._full(1.0), x2, self._full(r), d2, self._full(-r))
                x = self._lin(x, 1.0, self._noise(), self.hp["s_noise"] * up)
        return x

    # ---- linear multistep (k_diffusion.py:252-284)
    def _solve_lms(self, den, x, sig):
        from scipy import integrate
        order = self.hp["order"]
        t = np.asarray(sig, dtype=np.float32)
        ds: List[torch.Tensor] = []
        for i in range(len(sig) - 1):
            d = self._lin(x, 1.0 / sig[i], den(x, sig[i]), -1.0 / sig[i])
            ds.append(d)
            if len(ds) > order:
                ds.pop(0)
            cur = min(i + 1, order)

            def coeff(j):
                def fn(tau):
                    p = 1.0
                    for k in range(cur):
                        if k != j:
                            p *= (tau - t[i - k]) / (t[i - j] - t[i - k])
                    return p
                return integrate.quad(fn, t[i], t[i + 1], epsrel=1e-4)[0]
            for j, dj in enumerate(reversed(ds)):
                x = self._lin(x, 1.0, dj, coeff(j))
        return x

    # ---- DPM-Solver++(2S) ancestral (k_diffusion.py:513-544)
    def _solve_dpmpp_2s_a(self, den, x, sig):
        for i in range(len(sig) - 1):
            d0 = den(x, sig[i])
            down, up = _ancestral(sig[i], sig[i + 1], self.hp["eta"])
            if down == 0:
                x = self._euler_to(x, d0, sig[i], down)
            else:
                t, t_next = -math.log(sig[i]), -math.log(down)
                h = t_next - t
                s = t + 0.5 * h
                x2 = self._lin(x, math.exp(-s) / math.exp(-t), d0, -math.expm1(-h * 0.5))
                d2 = den(x2, math.exp(-s))
                x = self._lin(x, math.exp(-t_next) / math.exp(-t), d2, -math.expm1(-h))
            if sig[i + 1] > 0:
                x = self._lin(x, 1.0, self._noise(), self.hp["s_noise"] * up)
        return x

    # ---- DPM-Solver++(2M) (k_diffusion.py:589-612)
    def _solve_dpmpp_2m(self, den, x, sig):
        old = None
        for i in range(len(sig) - 1):
            d0 = den(x, sig[i])
            if sig[i + 1] == 0:
                t = -math.log(sig[i])
                # sigma_fn(t_next) / sigma_fn(t) = 0, -(expm1(-inf)) = 1: x = denoised
                x = self._lin(d0, 1.0)
            else:
                t, t_next = -math.log(sig[i]), -math.log(sig[i + 1])
                h = t_next - t
                ratio, k = math.exp(-t_next) / math.exp(-t), -math.expm1(-h)
                if old is None:
                    x = self._lin(x, ratio, d0, k)
                else:
                    r = (t - (-math.log(sig[i - 1]))) / h
                    x = self._lin(x, ratio, d0, k * (1 + 1 / (2 * r)), old, -k / (2 * r))
            old = d0
        return x

    def _brownian(self, x, sig):
        """The noise sampler the reference builds as BrownianTreeNoiseSampler(x, sigmas[sigmas > 0].min(), sigmas.max())."""
        smin, smax = float(min(s for s in sig if s > 0)), float(max(sig))
        if self.brownian is not None:
            return self.brownian(x, self._randn_like(x), smin, smax)
        from .brownian import BrownianTreeNoise
        return BrownianTreeNoise(x, smin, smax)

    def _randn_like(self, x):
        dev = x.device
        return lambda shape: self._randn(shape, dev)

    # ---- DPM-Solver++ SDE (k_diffusion.py:547-586); Brownian time = sigma itself (identity transform)
    def _solve_dpmpp_sde(self, den, x, sig, r: float = 0.5):
        ns = self._brownian(x, sig)
        eta, s_noise = self.hp["eta"], self.hp["s_noise"]
        sf, tf = (lambda t: math.exp(-t)), (lambda s: -math.log(s) if s > 0 else math.inf)
        for i in range(len(sig) - 1):
            d0 = den(x, sig[i])
            if sig[i + 1] == 0:
                x = self._euler_to(x, d0, sig[i], sig[i + 1])
                continue
            t, t_next = tf(sig[i]), tf(sig[i + 1])
            h = t_next - t
            s = t + h * r
            fac = 1 / (2 * r)
            sd, su = _ancestral(sf(t), sf(s), eta)
            s_ = tf(sd)
            # the Brownian query times as the reference forms them: exp(-t) of float32 tensors (k_diffusion.py:555-556,576,583)
            t32, tn32 = -_log(sig[i]), -_log(sig[i + 1])
            q_t, q_s, q_n = (float(np.exp(-v)) for v in (t32, t32 + (tn32 - t32) * f32(r), tn32))
            x2 = self._lin(x, sf(s_) / sf(t), d0, -math.expm1(t - s_), ns(q_t, q_s), s_noise * su)
            d2 = den(x2, sf(s))
            sd, su = _ancestral(sf(t), sf(t_next), eta)
            tn_ = tf(sd)
            k = -math.expm1(t - tn_)
            x = ops.lincomb4(x, self._full(sf(tn_) / sf(t)), d0, self._full(k * (1 - fac)), d2, self._full(k * fac),
                             ns(q_t, q_n), self._full(s_noise * su))
        return x

    # ---- DPM-Solver++(2M) SDE, midpoint (k_diffusion.py:615-657)
    def _solve_dpmpp_2m_sde(self, den, x, sig):
        ns = self._brownian(x, sig)
        eta, s_noise = self.hp["eta"], self.hp["s_noise"]
        old, h_last = None, None
        for i in range(len(sig) - 1):
            d0 = den(x, sig[i])
            if sig[i + 1] == 0:
                x = self._lin(d0, 1.0)
                h = None
            else:
                t, s = -math.log(sig[i]), -math.log(sig[i + 1])
                h = s - t
                eh = eta * h
                k = -math.expm1(-h - eh)
                if old is None:
                    x = self._lin(x, sig[i + 1] / sig[i] * math.exp(-eh), d0, k)
                else:
                    c = 0.5 * k / (h_last / h)
                    x = self._lin(x, sig[i + 1] / sig[i] * math.exp(-eh), d0, k + c, old, -c)
                if eta:
                    x = self._lin(x, 1.0, ns(sig[i], sig[i + 1]), sig[i + 1] * math.sqrt(-math.expm1(-2 * eh)) * s_noise)
            old, h_last = d0, h
        return x

    # ---- DPM-Solver++(3M) SDE (k_diffusion.py:660-707)
    def _solve_dpmpp_3m_sde(self, den, x, sig):
        ns = self._brownian(x, sig)
        eta, s_noise = self.hp["eta"], self.hp["s_noise"]
        d1_, d2_, h1, h2 = None, None, None, None
        for i in range(len(sig) - 1):
            d0 = den(x, sig[i])
            if sig[i + 1] == 0:
                x = self._lin(d0, 1.0)
                h = None
            else:
                t, s = -math.log(sig[i]), -math.log(sig[i + 1])
                h = s - t
                he = h * (eta + 1)
                a, k = math.exp(-he), -math.expm1(-he)
                if h2 is not None:
                    r0, r1 = h1 / h, h2 / h
                    phi2 = math.expm1(-he) / he + 1
                    phi3 = phi2 / he - 0.5
                    # d1_0 = (d0 - d1_)/r0, d1_1 = (d1_ - d2_)/r1, d1 = d1_0 + (d1_0 - d1_1) r0/(r0+r1), d2 = (d1_0 - d1_1)/(r0+r1)
                    # x += phi2*d1 - phi3*d2  as a combination of d0, d1_, d2_
                    u = phi2 * (1 + r0 / (r0 + r1)) - phi3 / (r0 + r1)       # coefficient of d1_0
                    v = -phi2 * r0 / (r0 + r1) + phi3 / (r0 + r1)            # coefficient of d1_1
                    c0, c1, c2 = u / r0, -u / r0 + v / r1, -v / r1
                    x = ops.lincomb4(x, self._full(a), d0, self._full(k + c0), d1_, self._full(c1), d2_, self._full(c2))
                elif h1 is not None:
                    phi2 = math.expm1(-he) / he + 1
                    c = phi2 / (h1 / h)
                    x = self._lin(x, a, d0, k + c, d1_, -c)
                else:
                    x = self._lin(x, a, d0, k)
                if eta:
                    x = self._lin(x, 1.0, ns(sig[i], sig[i + 1]), sig[i + 1] * math.sqrt(-math.expm1(-2 * h * eta)) * s_noise)
            d1_, d2_ = d0, d1_
            h1, h2 = h, h1
        return x
