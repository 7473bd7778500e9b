"""TEST INFRASTRUCTURE ONLY — loader for the *unmodified* reference (`/root/reference`).

Used by `oracle/make_golden.py` and by `-m "not gpu"` tests to (a) validate the oracle restatement and
(b) generate golden vectors.  `/root/reference` does not exist on the GPU box, so nothing that runs there
may call :func:`load_reference`; use :func:`have_reference` to gate.

Shims: stub modules under `oracle/refshim` for omegaconf / timm / ftfy / torchsde,
`torch.Tuple` (reference sampler/edm_sampler.py:145) and a no-op `torch.cuda.synchronize` on a GPU-less host
(reference utils/common.py:271).
"""
import os
import sys
import types
import typing

REFERENCE_ROOT = os.environ.get("DIFFBIR_REFERENCE", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")


def have_reference() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "diffbir"))


def load_reference() -> types.SimpleNamespace:
    """Import the reference's model / sampler / pipeline modules (CPU). Returns a namespace."""
    if not have_reference():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    import torch

    for p in (REFERENCE_ROOT, _SHIM):
        if p in sys.path:
            sys.path.remove(p)
    sys.path[:0] = [_SHIM, REFERENCE_ROOT]
    if not hasattr(torch, "Tuple"):
        torch.Tuple = typing.Tuple
    if not torch.cuda.is_available():
        torch.cuda.synchronize = lambda *a, **k: None
    import importlib

    # The reference's `diffbir/` has no __init__.py (namespace package), so the engine's `diffbir` alias package at the
    # repo root would win regardless of sys.path order: bind the name to the reference directory explicitly (and drop
    # any alias modules already imported in this process).
    for k in [k for k in sys.modules if k == "diffbir" or k.startswith("diffbir.")]:
        del sys.modules[k]
    pkg = types.ModuleType("diffbir")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "diffbir")]
    sys.modules["diffbir"] = pkg

    model = importlib.import_module("diffbir.model")
    pipeline = importlib.import_module("diffbir.pipeline")
    sampler = importlib.import_module("diffbir.sampler")
    common = importlib.import_module("diffbir.utils.common")
    return types.SimpleNamespace(
        model=model, pipeline=pipeline, sampler=sampler, common=common,
        ControlLDM=model.ControlLDM, SwinIR=model.SwinIR, Diffusion=model.Diffusion,
        SwinIRPipeline=pipeline.SwinIRPipeline,
        SpacedSampler=sampler.SpacedSampler, DPMSolverSampler=sampler.DPMSolverSampler,
    )
