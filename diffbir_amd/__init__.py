"""diffbir_amd — MI355X-native DiffBIR inference hot path (HIP kernels behind a C ABI)."""
__version__ = "0.1.0"

import os as _os

# Launch path: kernel arguments in device memory.  It is the HIP runtime's own default on MI355X / ROCm 7.2; it is pinned
# here because the engine issues ~700 launches per network evaluation and an inherited HIP_FORCE_DEV_KERNARG=0 costs 2.5 - 3 %
# of the C2 benchmark (profiles/r4_kernarg_ab.txt).  The runtime reads the variable when it initialises (first HIP call of the
# process); an explicit value in the environment wins.  INTEGRATION.md, "Runtime environment".
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
