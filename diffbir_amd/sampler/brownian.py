"""The Brownian motion of the SDE solvers (`edm_dpm++_sde`, `_2m_sde`, `_3m_sde`): a restatement of what the reference gets
from `torchsde.BrownianTree` through k-diffusion's wrappers (reference sampler/k_diffusion.py:70-119; torchsde==0.2.6 is pinned
in the reference's requirements.txt:22 and is NOT vendored in /root/reference nor installable here).

PARITY UNPINNED against torchsde itself: no torchsde build exists in this environment, so the tree below follows the published
algorithm of torchsde 0.2.x's `BrownianInterval` (Kidger et al. 2021, "Efficient and Accurate Gradients for Neural SDEs", §4 —
the "Brownian Interval") as restated here from its description, and is pinned by (i) the process invariants (additivity,
variance, independence, determinism per seed, independence of the query order: tests/test_brownian_cpu.py), (ii) a second,
independently structured restatement in oracle/refshim/torchsde that the UNMODIFIED reference wrappers
(BatchedBrownianTree / BrownianTreeNoiseSampler) run on when the goldens are generated.  Seed-for-seed equality with a real
torchsde install is therefore plausible, not verified (INTEGRATION.md says so).

The algorithm (W only: k-diffusion asks for increments, never Levy areas):
  * the top interval [t0, t1] holds W(t0, t1) = sqrt(t1 - t0) * N(seed_W), seed_W = the first word of
    SeedSequence(entropy, pool_size).generate_state(2);
  * every interval is split at its (rounded) midpoint ("halfway tree"; query times are rounded to -log10(tol) digits, so the
    bisection ends on the query grid after ~log2((t1 - t0) / tol) levels); a node at depth d with integer key k (left child 2k,
    right child 2k + 1) draws its split noise from SeedSequence(entropy, spawn_key=(k, d), pool_size).generate_state(4)[0];
  * the left child's increment is the Brownian bridge  (l / h) W + sqrt(l r / h) N(seed)  (l, r = child lengths, h = l + r), the
    right child's is W minus it;
  * a query [ta, tb] is the ordered list of tree nodes that tile it; the search starts at the node the previous query ended on;
  * noise tensors come from `torch.randn(size, generator=torch.Generator(device).manual_seed(seed))` — torch's generator on
    the tensor's own device, as the reference draws it — and a bounded cache only saves recomputation (values are a function
    of the seed).
"""
import math
from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch


class _Node:
    __slots__ = ("start", "end", "parent", "is_left", "key", "depth", "mid", "seed", "left", "right")

    def __init__(self, start, end, parent, is_left, key, depth):
        self.start, self.end, self.parent, self.is_left, self.key, self.depth = start, end, parent, is_left, key, depth
        self.mid = None
        self.seed = self.left = self.right = None


class BrownianTree:
    """W on [t0, t1] for one entropy value; `tree(ta, tb)` = W(tb) - W(ta) (f32 tensor of `shape` on `device`)."""

    def __init__(self, t0: float, t1: float, shape: Sequence[int], entropy: int, device, dtype=torch.float32,
                 tol: float = 1e-6, pool_size: int = 24, cache_size: int = 45, noise_device=None):
        t0, t1 = float(t0), float(t1)
        if not t0 < t1:
            raise ValueError(f"need t0 < t1, got {t0}, {t1}")
        if tol <= 0:
            raise ValueError("tol must be positive")
        self.shape, self.device, self.dtype = tuple(shape), torch.device(device), dtype
        # where the generator lives: the reference draws on the tensor's device; tests compare a GPU run with goldens the
        # reference produced on CPU and pin the draws to the CPU generator
        self.noise_device = torch.device(noise_device) if noise_device is not None else self.device
        self.entropy, self.pool_size = int(entropy), int(pool_size)
        self._digits = -int(math.log10(tol))
        self.root = _Node(self._round(t0), self._round(t1), None, None, 0, 0)
        seed_w = np.random.SeedSequence(entropy=self.entropy, pool_size=self.pool_size).generate_state(2)[0]
        self._w_root = self._randn(seed_w) * math.sqrt(t1 - t0)
        self._cache: "dict[_Node, torch.Tensor]" = {}
        self._order: List[_Node] = []
        self._cache_size = cache_size
        self._last = self.root

    def _round(self, t: float) -> float:
        return round(t, self._digits)

    def _randn(self, seed) -> torch.Tensor:
        g = torch.Generator(self.noise_device).manual_seed(int(seed))
        w = torch.randn(self.shape, dtype=self.dtype, device=self.noise_device, generator=g)
        return w if self.noise_device == self.device else w.to(self.device)

    # ---- tree
    def _split(self, n: _Node, at: float) -> None:
        """Bisect `n` (and then the half that holds `at`) until `at` is a node boundary."""
        while True:
            n.mid = self._round(0.5 * (n.end + n.start))
            n.seed = np.random.SeedSequence(entropy=self.entropy, spawn_key=(n.key, n.depth),
                                            pool_size=self.pool_size).generate_state(4)[0]
            n.left = _Node(n.start, n.mid, n, True, 2 * n.key, n.depth + 1)
            n.right = _Node(n.mid, n.end, n, False, 2 * n.key + 1, n.depth + 1)
            if at > n.mid:
                n = n.right
            elif at < n.mid:
                n = n.left
            else:
                return

    def _locate(self, ta: float, tb: float) -> List[_Node]:
        out: List[_Node] = []
        todo = [(self._last, ta, tb)]
        while todo:
            n, a, b = todo.pop()
            while True:
                if a < n.start or b > n.end:
                    n = n.parent
                elif a == n.start and b == n.end:
                    out.append(n)
                    break
                elif n.mid is None:
                    if a == n.start:
                        self._split(n, b)
                        n = n.left
                    else:
                        self._split(n, a)
                        n = n.right
                elif b <= n.mid:
                    n = n.left
                elif a >= n.mid:
                    n = n.right
                else:   # straddles the midpoint: left part first (order of `out`), then the right part
                    todo.append((n.right, n.mid, b))
                    b = n.mid
                    n = n.left
        return out

    def _w(self, n: _Node) -> torch.Tensor:
        """Increment over node `n`: walk up to the nearest cached ancestor, then bridge down."""
        path = []
        while n.parent is not None and n not in self._cache:
            path.append(n)
            n = n.parent
        w = self._w_root if n.parent is None else self._cache[n]
        for c in reversed(path):
            p = c.parent
            hr = 1 / (p.end - p.start)
            lo, hi = p.mid - p.start, p.end - p.mid
            left = lo * w * hr + math.sqrt(lo * hi * hr) * self._randn(p.seed)   # bridge mean + sd * N(0, 1)
            w = left if c.is_left else w - left
            self._remember(c, w)
        return w

    def _remember(self, n: _Node, w: torch.Tensor) -> None:
        if self._cache_size <= 0:
            return
        if len(self._order) >= self._cache_size:
            self._cache.pop(self._order.pop(0), None)
        self._cache[n] = w
        self._order.append(n)

    def __call__(self, ta: float, tb: float) -> torch.Tensor:
        ta = min(max(float(ta), self.root.start), self.root.end)   # torchsde clamps (with a warning) instead of raising
        tb = min(max(float(tb), self.root.start), self.root.end)
        if ta > tb:
            raise RuntimeError(f"Query times ta={ta:.3f} and tb={tb:.3f} must respect ta <= tb.")
        ta, tb = self._round(ta), self._round(tb)
        if ta == tb:
            return torch.zeros(self.shape, dtype=self.dtype, device=self.device)
        nodes = self._locate(ta, tb)
        self._last = nodes[-1]
        w = self._w(nodes[0])
        for n in nodes[1:]:
            w = w + self._w(n)
        return w


def draw_seed() -> int:
    """The seed k-diffusion draws when none is given (k_diffusion.py:78-79): one int64 from torch's GLOBAL CPU generator."""
    return int(torch.randint(0, 2 ** 63 - 1, []).item())


class BrownianTreeNoise:
    """k-diffusion's BrownianTreeNoiseSampler over BatchedBrownianTree (k_diffusion.py:70-119):  noise(sigma, sigma_next) =
    W(t0, t1) / sqrt(|t1 - t0|), t = transform(sigma), sign-corrected for descending queries; `seed` an int (one tree of the
    whole batch's shape), a list of ints (one tree per batch item, stacked) or None (drawn from the global CPU generator)."""

    def __init__(self, x: torch.Tensor, sigma_min: float, sigma_max: float, seed: Union[None, int, Sequence[int]] = None,
                 transform: Callable[[float], float] = lambda s: s, noise_device=None):
        self.transform = transform
        t0, t1, self.sign = self._sort(float(transform(float(sigma_min))), float(transform(float(sigma_max))))
        if seed is None:
            seed = draw_seed()
        try:
            seeds = [int(s) for s in seed]
            if len(seeds) != x.shape[0]:
                raise AssertionError("one seed per batch item")
            shape, self.batched = tuple(x.shape[1:]), True
        except TypeError:
            seeds, shape, self.batched = [int(seed)], tuple(x.shape), False
        self.trees = [BrownianTree(t0, t1, shape, s, x.device, torch.float32, noise_device=noise_device) for s in seeds]

    @staticmethod
    def _sort(a, b):
        return (a, b, 1) if a < b else (b, a, -1)

    def __call__(self, sigma: float, sigma_next: float) -> torch.Tensor:
        ta, tb = float(self.transform(float(sigma))), float(self.transform(float(sigma_next)))
        t0, t1, sign = self._sort(ta, tb)
        w = torch.stack([tree(t0, t1) for tree in self.trees]) * (self.sign * sign)
        w = w if self.batched else w[0]
        return w / math.sqrt(abs(tb - ta))
