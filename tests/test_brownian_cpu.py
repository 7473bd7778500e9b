"""The Brownian tree of the SDE solvers (diffbir_amd/sampler/brownian.py), a restatement of torchsde's BrownianTree as the
reference uses it (k_diffusion.py:70-119; torchsde is not installable here, so parity with it is unpinned — see the module
header).  What CAN be pinned without torchsde: the process invariants, determinism, and agreement with the second,
differently structured restatement the golden generator runs the reference on (oracle/refshim/torchsde)."""
import math
import os
import random
import sys

import numpy as np
import pytest
import torch

from diffbir_amd.sampler.brownian import BrownianTree, BrownianTreeNoise

SHIM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "refshim")


def _shim():
    sys.path.insert(0, SHIM)
    try:
        import importlib
        sys.modules.pop("torchsde", None)
        return importlib.import_module("torchsde")
    finally:
        sys.path.remove(SHIM)


def test_tree_equals_the_oracle_restatement_on_random_queries():
    """Two independently structured restatements (iterative node tree that resumes at the last node / pure recursion from the
    root) give the same increments for 200 random queries over the samplers' sigma range, in any order."""
    sde = _shim()
    x = torch.zeros(2, 4, 8, 8)
    rnd = random.Random(3)
    for entropy in (0, 1234, 2 ** 63 - 2):
        t = BrownianTree(0.0292, 1e4, x.shape, entropy, "cpu")
        r = sde.BrownianTree(0.0292, x, 1e4, entropy=entropy)
        for i in range(70):
            lo, hi = ((0.0292, 1e4), (0.0292, 1.0), (0.5, 30.0))[i % 3]
            a, b = sorted(rnd.uniform(lo, hi) for _ in range(2))
            wa, wb = t(a, b), r(a, b)
            assert torch.allclose(wa, wb, atol=2e-6 * max(1.0, math.sqrt(b - a)), rtol=1e-5), (entropy, a, b)


def test_increments_add_up_and_do_not_depend_on_the_query_order():
    shape = (1, 4, 16, 16)
    pts = [1e4, 312.5, 77.0, 14.2, 3.3, 0.9, 0.21, 0.0292]
    a = BrownianTree(0.0292, 1e4, shape, 99, "cpu")
    fwd = [a(pts[i + 1], pts[i]) for i in range(len(pts) - 1)]                  # descending sigma, as the solvers ask
    b = BrownianTree(0.0292, 1e4, shape, 99, "cpu")
    rev = [b(pts[i + 1], pts[i]) for i in reversed(range(len(pts) - 1))][::-1]  # a fresh tree, the other way round
    for u, v in zip(fwd, rev):
        assert torch.allclose(u, v, atol=1e-5)
    total = a(pts[-1], pts[0])
    assert torch.allclose(sum(fwd), total, atol=1e-3)                            # |W| ~ 100 here
    assert torch.equal(a(pts[3], pts[2]), a(pts[3], pts[2]))                     # revisiting = the same value
    tiny = BrownianTree(0.0292, 1e4, shape, 99, "cpu", cache_size=0)             # the cache only saves recomputation
    assert torch.allclose(tiny(pts[3], pts[2]), a(pts[3], pts[2]), atol=1e-6)
    other = BrownianTree(0.0292, 1e4, shape, 100, "cpu")
    assert not torch.allclose(other(pts[3], pts[2]), a(pts[3], pts[2]), atol=1e-2)


def test_increments_are_gaussian_with_variance_dt_and_independent():
    shape = (4, 4, 64, 64)                                                      # 65536 samples per increment
    t = BrownianTree(0.0292, 1e4, shape, 7, "cpu")
    spans = [(0.0292, 0.5), (0.5, 3.0), (3.0, 40.0), (40.0, 900.0), (900.0, 1e4), (1.0, 1.000002)]
    w = [t(a, b) / math.sqrt(round(b, 6) - round(a, 6)) for a, b in spans]
    for v in w:
        assert abs(v.mean().item()) < 0.02 and abs(v.var().item() - 1.0) < 0.03
        assert abs((v ** 4).mean().item() - 3.0) < 0.15                          # Gaussian fourth moment
    for i in range(4):                                                           # disjoint intervals: uncorrelated
        for j in range(i + 1, 5):
            assert abs((w[i] * w[j]).mean().item()) < 0.02
    # nested: W over a sub-interval against the rest of its parent interval
    inner, outer = t(3.0, 10.0), t(3.0, 40.0)
    assert abs(((outer - inner) * inner).mean().item()) / math.sqrt(7 * 30) < 0.02


def test_noise_sampler_mirrors_the_reference_wrappers():
    """BrownianTreeNoise == k-diffusion's BrownianTreeNoiseSampler + BatchedBrownianTree (k_diffusion.py:70-119) on the
    restated tree: scalar seed, per-item seeds, seed from the global CPU generator, descending queries, a transform."""
    ref_import = pytest.importorskip("oracle.ref_import")
    if not ref_import.have_reference():
        pytest.skip("reference checkout not present")
    ref_import.load_reference()
    import importlib
    kd = importlib.import_module("diffbir.sampler.k_diffusion")
    if kd.BrownianTreeNoiseSampler.__module__ != kd.__name__:
        pytest.skip("k_diffusion was patched by an earlier golden-generator import")
    x = torch.zeros(3, 4, 8, 8)
    smin, smax = torch.tensor(0.0292), torch.tensor(1e4)
    for seed in (5, [11, 12, 13]):
        mine, ref = BrownianTreeNoise(x, smin.item(), smax.item(), seed), kd.BrownianTreeNoiseSampler(x, smin, smax, seed)
        for s0, s1 in ((1e4, 312.5), (312.5, 14.0), (14.0, 0.0292), (0.5, 0.7)):
            u, v = mine(s0, s1), ref(torch.tensor(s0), torch.tensor(s1))
            assert u.shape == v.shape == x.shape and torch.allclose(u, v, atol=1e-5), (seed, s0, s1)
    torch.manual_seed(3)
    ref = kd.BrownianTreeNoiseSampler(x, smin, smax)
    torch.manual_seed(3)
    mine = BrownianTreeNoise(x, smin.item(), smax.item())
    assert torch.allclose(mine(50.0, 2.0), ref(torch.tensor(50.0), torch.tensor(2.0)), atol=1e-5)
    tf = lambda s: -math.log(s)                                                 # noqa: E731 - log-sigma time
    mine = BrownianTreeNoise(x, smin.item(), smax.item(), 9, transform=tf)
    ref = kd.BrownianTreeNoiseSampler(x, smin, smax, 9, transform=lambda s: s.log().neg())
    assert torch.allclose(mine(50.0, 2.0), ref(torch.tensor(50.0), torch.tensor(2.0)), atol=1e-5)


def test_argument_checks():
    with pytest.raises(ValueError):
        BrownianTree(1.0, 1.0, (2,), 0, "cpu")
    t = BrownianTree(0.0, 1.0, (2,), 0, "cpu")
    with pytest.raises(RuntimeError):
        t(0.7, 0.2)
    assert torch.equal(t(0.3, 0.3), torch.zeros(2))
    assert torch.equal(t(-5.0, 0.25), t(0.0, 0.25)) and torch.equal(t(0.25, 9.0), t(0.25, 1.0))   # clamped like torchsde
    assert np.isfinite(t(0.123456, 0.123457).numpy()).all()                      # one grid step
    with pytest.raises(AssertionError):
        BrownianTreeNoise(torch.zeros(3, 2), 0.1, 1.0, seed=[1, 2])
