"""Host-side tables of the path — timestep spacing, spaced-sampler buffers, the DPM-Solver discrete VP schedule, tile
windows and Gaussian blend weights — of BOTH the oracle and the engine against tables produced by the reference's own
functions (tests/golden/host_tables.json, oracle/make_golden.py host_tables).  Integer tables must match exactly,
float tables to float32 rounding."""
import json
import os

import numpy as np
import pytest
import torch

from diffbir_amd import configs
from diffbir_amd.model import Diffusion
from diffbir_amd.sampler import spaced_sampler as eng_sp
from diffbir_amd.sampler.dpms_sampler import NoiseScheduleVP
from diffbir_amd.utils import common as eng_common
from oracle import sampling as orc


@pytest.fixture(scope="module")
def tables(golden_dir):
    with open(os.path.join(golden_dir, "host_tables.json")) as f:
        return json.load(f)


def test_space_timesteps_exact(tables):
    for key, ref in tables["space_timesteps"].items():
        got = sorted(eng_sp.space_timesteps(1000, key))
        assert got == ref, key
        if key.isdigit():
            assert sorted(orc.space_timesteps(1000, int(key))) == ref, key


@pytest.mark.parametrize("name", ["DIFFUSION_V2", "DIFFUSION_V21"])
@pytest.mark.parametrize("steps", [5, 50])
def test_spaced_sampler_tables(tables, name, steps):
    ref = tables["spaced_tables"][f"{name}_{steps}"]
    diff = Diffusion(**configs.get(name))
    smp = eng_sp.SpacedSampler(diff.betas, diff.parameterization, rescale_cfg=False)
    smp.make_schedule(steps)
    assert [int(t) for t in smp.timesteps] == ref["timesteps"]
    otb = orc.spaced_tables(orc.make_betas(**configs.get(name)), steps)
    for k, v in ref.items():
        if k == "timesteps":
            continue
        want = np.array([np.nan if x is None else x for x in v], dtype=np.float64)
        fin = np.isfinite(want)
        got = smp.tables[k].double().numpy()
        assert got.shape == want.shape and np.allclose(got[fin], want[fin], rtol=1e-6, atol=1e-9), k
        if k in otb:
            o = np.asarray(otb[k], dtype=np.float32).astype(np.float64)
            assert np.allclose(o[fin], want[fin], rtol=1e-6, atol=1e-9), k


@pytest.mark.parametrize("name", ["DIFFUSION_V2", "DIFFUSION_V21"])
def test_dpm_solver_schedule(tables, name):
    """NoiseScheduleVP incl. numerical_clip_alpha (total_N = 986 for the zero-terminal-SNR schedule) and
    the 20-step time_uniform grid fed to the network as (t - 1/N) * 1000."""
    ref = tables["dpm"][name]
    diff = Diffusion(**configs.get(name))
    ns = NoiseScheduleVP(torch.tensor(diff.betas, dtype=torch.float32))
    assert ns.total_N == ref["total_N"]
    ts = torch.linspace(ns.T, 1.0 / ns.total_N, 21)
    assert np.allclose(ts.numpy(), ref["t"], rtol=1e-6, atol=1e-7)
    for key, fn in (("alpha", ns.marginal_alpha), ("std", ns.marginal_std), ("lam", ns.marginal_lambda)):
        assert np.allclose(fn(ts).numpy(), ref[key], rtol=2e-5, atol=2e-6), key
    assert np.allclose(((ts - 1.0 / ns.total_N) * 1000.0).numpy(), ref["model_t"], rtol=1e-6, atol=1e-4)
    o = orc.VPSchedule(orc.make_betas(**configs.get(name)))
    assert o.total_N == ref["total_N"]
    for key, fn in (("alpha", o.alpha), ("std", o.std), ("lam", o.lam)):
        assert np.allclose(fn(ts).numpy(), ref[key], rtol=2e-5, atol=2e-6), ("oracle", key)


def test_sliding_windows_exact(tables):
    for key, ref in tables["sliding_windows"].items():
        hw, size, stride = key.split("_")
        h, w = (int(x) for x in hw.split("x"))
        assert [list(x) for x in eng_common.sliding_windows(h, w, int(size), int(stride))] == ref, key
        assert [list(x) for x in orc.sliding_windows(h, w, int(size), int(stride))] == ref, key


def test_gaussian_weights(tables):
    for key, ref in tables["gaussian_weights"].items():
        tw, th = (int(x) for x in key.split("x"))
        for fn in (eng_common.gaussian_weights, orc.gaussian_weights):
            got = np.asarray(fn(tw, th), dtype=np.float64)
            if "full" in ref:
                assert np.allclose(got, np.array(ref["full"]), rtol=1e-6, atol=0), key
            else:
                assert list(got.shape) == ref["shape"] and np.isclose(got.sum(), ref["total"], rtol=1e-6)
                for r, row in ref["rows"].items():
                    assert np.allclose(got[int(r)], row, rtol=1e-6, atol=0), (key, r)
                for c, col in ref["cols"].items():
                    assert np.allclose(got[:, int(c)], col, rtol=1e-6, atol=0), (key, c)
