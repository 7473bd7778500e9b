"""Reference `diffbir.inference` surface: the loops whose stage-1 cleaner is SwinIR (BSR v1 / v2.1, aligned BFR).
BIDInferenceLoop (SCUNet), UnAlignedBFRInferenceLoop (RetinaFace) and CustomInferenceLoop are separate products outside
this engine's scope."""
from .bfr_loop import BFRInferenceLoop  # noqa: F401
from .bsr_loop import BSRInferenceLoop  # noqa: F401
from .loop import EmptyCaptioner, InferenceLoop, load_config  # noqa: F401
from .pretrained_models import MODELS  # noqa: F401
