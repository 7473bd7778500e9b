"""NativeModule — minimal stand-in for nn.Module on the drop-in boundary.

The reference's callers use only: construction from YAML params, ``load_state_dict(sd, strict)``,
``state_dict()``, ``.eval()``, ``.to(device)``, ``.parameters()``, ``__call__``.  The engine keeps checkpoints'
key names (model/specs.py) but not the reference's module classes: weights live in a flat dict and are
re-packed for the MFMA kernels (`_pack`) the first time the module runs on a device.
"""
from collections import OrderedDict
from typing import Dict, Iterable, Optional

import torch


class NativeModule:
    def __init__(self, spec: "OrderedDict[str, tuple]"):
        self._spec = spec
        self._sd: Dict[str, torch.Tensor] = {}
        self._device = torch.device("cpu")
        self._dtype = torch.float16
        self._packed = False
        self._gen = 0  # bumped whenever the packed weights are rebuilt (captured HIP graphs key on it)
        self.training = False

    # ---- nn.Module-like surface -------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        mine = {k for k, (_, kind) in self._spec.items() if kind != "buf"}
        bufs = {k for k, (_, kind) in self._spec.items() if kind == "buf"}
        given = set(sd.keys())
        missing = sorted(mine - given)
        unexpected = sorted(given - mine - bufs)
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing[:8]}"
                               f"{'...' if len(missing) > 8 else ''}, unexpected keys {unexpected[:8]}")
        for k in mine & given:
            shp = tuple(self._spec[k][0])
            v = sd[k]
            if tuple(v.shape) != shp:
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {shp}")
            self._sd[k] = v.detach()  # kept where it lives (CPU checkpoints stay on CPU); cast when packed
        self._packed = False
        return missing, unexpected

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, self._sd[k]) for k in self._spec if k in self._sd)

    def parameters(self) -> Iterable[torch.Tensor]:
        return iter(self._sd.values())

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("diffbir_amd is an inference engine")
        return self

    def to(self, device=None, dtype: Optional[torch.dtype] = None):
        if device is not None and not isinstance(device, torch.dtype):
            d = torch.device(device)
            if d != self._device:
                self._device, self._packed = d, False
        if isinstance(device, torch.dtype):
            dtype = device
        if dtype is not None and dtype != self._dtype:
            self.set_dtype(dtype)
        return self

    def set_dtype(self, dtype: torch.dtype):
        if dtype not in (torch.float16, torch.bfloat16):
            raise NotImplementedError("the MFMA engine computes in float16 or bfloat16 (f32 accumulate); "
                                      f"got {dtype}")
        if dtype != self._dtype:
            self._dtype, self._packed = dtype, False
        return self

    @property
    def device(self):
        return self._device

    # ---- packing --------------------------------------------------------------------------------
    def _ensure_packed(self):
        if not self._packed:
            need = [k for k, (_, kind) in self._spec.items() if kind != "buf" and k not in self._sd]
            if need:
                raise RuntimeError(f"{type(self).__name__}: weights not loaded (e.g. {need[:3]})")
            self._pack()
            self._packed = True
            self._gen += 1

    def _pack(self):  # pragma: no cover - abstract
        raise NotImplementedError

    def _w(self, key: str) -> torch.Tensor:
        return self._sd[key]

    def _f32(self, key: str) -> torch.Tensor:
        return self._sd[key].to(self._device, torch.float32).contiguous()

    def release_master(self):
        """Drop the unpacked checkpoint tensors once packed (frees ~4 bytes/param of host or device memory)."""
        self._ensure_packed()
        self._sd = {}
