from .spaced_sampler import SpacedSampler, space_timesteps  # noqa: F401
from .dpms_sampler import DPMSolverSampler  # noqa: F401
from .sampler import Sampler  # noqa: F401
from .ddim_sampler import DDIMSampler  # noqa: F401
from .edm_sampler import EDMSampler  # noqa: F401
