"""Reference `diffbir.inference` surface: BSR (SwinIR / BSRNet), aligned BFR (SwinIR), BID (SwinIR / SCUNet) and custom
(self-trained weights) loops.  UnAlignedBFRInferenceLoop (RetinaFace face detection front-end) is a separate product
outside this engine's scope."""
from .bfr_loop import BFRInferenceLoop  # noqa: F401
from .bid_loop import BIDInferenceLoop  # noqa: F401
from .bsr_loop import BSRInferenceLoop  # noqa: F401
from .custom_loop import CustomInferenceLoop  # noqa: F401
from .loop import EmptyCaptioner, InferenceLoop, load_config  # noqa: F401
from .pretrained_models import MODELS  # noqa: F401
