"""Reference `diffbir.inference` surface: BSR (SwinIR / BSRNet), aligned BFR (SwinIR) and BID (SwinIR / SCUNet) loops.
UnAlignedBFRInferenceLoop (RetinaFace face detection front-end) and CustomInferenceLoop are separate products outside
this engine's scope."""
from .bfr_loop import BFRInferenceLoop  # noqa: F401
from .bid_loop import BIDInferenceLoop  # noqa: F401
from .bsr_loop import BSRInferenceLoop  # noqa: F401
from .loop import EmptyCaptioner, InferenceLoop, load_config  # noqa: F401
from .pretrained_models import MODELS  # noqa: F401
