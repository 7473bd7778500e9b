from .cldm import ControlLDM  # noqa: F401
from .swinir import SwinIR  # noqa: F401
from .gaussian_diffusion import Diffusion  # noqa: F401
from .unet import ControlledUnetModel, ControlNet  # noqa: F401
from .vae import AutoencoderKL  # noqa: F401
from .clip import FrozenOpenCLIPEmbedder  # noqa: F401
from .bsrnet import RRDBNet  # noqa: F401
from .scunet import SCUNet  # noqa: F401
