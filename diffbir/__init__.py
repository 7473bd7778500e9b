"""`diffbir` — import-path alias of the MI355X engine (`diffbir_amd`), so that code written against the reference
package runs unchanged: `from diffbir.pipeline import SwinIRPipeline`, `from diffbir.model import ControlLDM, SwinIR,
Diffusion`, `from diffbir.sampler import SpacedSampler`, `from diffbir.inference import BSRInferenceLoop`,
`from diffbir.utils.common import instantiate_from_config` (reference diffbir/__init__.py is empty; its sub-packages are
what callers import — SURVEY.md §8b).  Every `diffbir.X` module object IS the `diffbir_amd.X` module (one copy of the
engine state: native library handle, tuning table, caches)."""
import importlib
import pkgutil
import sys

import diffbir_amd as _impl

__version__ = _impl.__version__
for _m in pkgutil.walk_packages(_impl.__path__, "diffbir_amd."):
    if _m.name.rsplit(".", 1)[-1].startswith("lib"):   # the in-tree HIP shared library is not a Python module
        continue
    _mod = importlib.import_module(_m.name)
    _alias = "diffbir" + _m.name[len("diffbir_amd"):]
    sys.modules[_alias] = _mod
    if _alias.count(".") == 1:
        globals()[_alias.split(".")[1]] = _mod
del _m, _mod, _alias
