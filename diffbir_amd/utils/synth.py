"""Deterministic synthetic weights (no checkpoints or network in the build/bench environment).

Every tensor is drawn from its own CPU generator seeded by ``crc32(key) ^ seed`` so that the same key
gives the same values regardless of construction order, process or machine — the oracle, the reference
(via load_state_dict) and the engine can therefore be fed bit-identical weights without shipping them.

The reference zero-initialises several layers (zero_module: unet.py:177-179, attention.py:331, unet.py:678,
controlnet.py:309-312) which would make every parity test vacuous, so *all* dense weights
are drawn ~ N(0, gain^2 / fan_in).
"""
import zlib
from typing import Dict

import torch


def synth_tensor(key: str, shape, kind: str, seed: int = 0, gain: float = 1.0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(shape)
    if kind == "w":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g) * (gain / max(fan_in, 1) ** 0.5)
    if kind == "b":
        return torch.randn(shape, generator=g) * 0.02
    if kind == "g":
        return 1.0 + torch.randn(shape, generator=g) * 0.05
    if kind == "e":
        return torch.randn(shape, generator=g) * 0.02
    raise ValueError(kind)


def synth_state_dict(spec, seed: int = 0, prefix: str = "", gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Materialise a spec (model/specs.py) into a state dict; "buf" entries are skipped (recomputed)."""
    return {k: synth_tensor(prefix + k, shp, kind, seed, gain)
            for k, (shp, kind) in spec.items() if kind != "buf"}
