"""Stand-in for torchsde (pinned torchsde==0.2.6 in the reference's requirements.txt:22; absent from /root/reference and not
installable here) so that the reference's own `BatchedBrownianTree` / `BrownianTreeNoiseSampler`
(/root/reference/diffbir/sampler/k_diffusion.py:70-119) run UNMODIFIED when oracle/make_golden.py generates the SDE-sampler goldens.

TEST INFRASTRUCTURE, PARITY UNPINNED: this is a restatement of the published "Brownian Interval" algorithm (Kidger et al. 2021,
§4) as torchsde 0.2.x implements it for `BrownianTree` (halfway tree, query times rounded to -log10(tol) digits, numpy
SeedSequence keyed by tree position, torch.randn from a per-node seeded generator).  No torchsde install exists to check it
against.  It is written as a pure recursion from the root, deliberately structured differently from the engine's iterative
node tree (diffbir_amd/sampler/brownian.py); tests/test_brownian_cpu.py compares the two and checks the process invariants.
Only `BrownianTree` is provided (the one name the reference uses)."""
import math

import numpy as np
import torch

__version__ = "0.2.6-restated"


class BrownianTree:
    def __init__(self, t0, w0, t1=None, W1=None, entropy=None, tol=1e-6, pool_size=24, cache_depth=24, safety=None):
        self.t0 = float(t0)
        self.t1 = self.t0 + 1.0 if t1 is None else float(t1)
        if not self.t0 < self.t1:
            raise ValueError("t0 must be strictly less than t1")
        self.w0 = w0
        self.entropy = int(np.random.randint(0, 2 ** 31 - 1)) if entropy is None else int(entropy)
        self.pool = pool_size
        self.nd = -int(math.log10(tol))
        if W1 is None:
            seed = np.random.SeedSequence(entropy=self.entropy, pool_size=self.pool).generate_state(2)[0]
            self.W = self._noise(seed) * math.sqrt(self.t1 - self.t0)
        else:
            self.W = W1 - w0
        self.memo = {}

    def _noise(self, seed):
        gen = torch.Generator(self.w0.device).manual_seed(int(seed))
        return torch.randn(self.w0.shape, dtype=self.w0.dtype, device=self.w0.device, generator=gen)

    def _halves(self, key, depth, s, e, w):
        """(midpoint, increment over [s, m], increment over [m, e]) of the node (key, depth) whose increment is w."""
        hit = self.memo.get((key, depth))
        if hit is None:
            m = round(0.5 * (e + s), self.nd)
            seed = np.random.SeedSequence(entropy=self.entropy, spawn_key=(key, depth), pool_size=self.pool).generate_state(4)[0]
            rh = 1 / (e - s)
            wl = (m - s) * w * rh + math.sqrt((m - s) * (e - m) * rh) * self._noise(seed)
            hit = (m, wl, w - wl)
            if len(self.memo) > 4096:
                self.memo.clear()
            self.memo[(key, depth)] = hit
        return hit

    def _between(self, key, depth, s, e, w, a, b):
        if a == s and b == e:
            return w
        m, wl, wr = self._halves(key, depth, s, e, w)
        if b <= m:
            return self._between(2 * key, depth + 1, s, m, wl, a, b)
        if a >= m:
            return self._between(2 * key + 1, depth + 1, m, e, wr, a, b)
        return self._between(2 * key, depth + 1, s, m, wl, a, m) + self._between(2 * key + 1, depth + 1, m, e, wr, m, b)

    def __call__(self, ta, tb=None):
        if tb is None:
            return self.w0 + self(self.t0, ta)
        s, e = round(self.t0, self.nd), round(self.t1, self.nd)
        ta, tb = min(max(float(ta), s), e), min(max(float(tb), s), e)
        if ta > tb:
            raise RuntimeError(f"Query times ta={ta:.3f} and tb={tb:.3f} must respect ta <= tb.")
        ta, tb = round(ta, self.nd), round(tb, self.nd)
        if ta == tb:
            return torch.zeros_like(self.w0)
        return self._between(0, 0, s, e, self.W, ta, tb)
