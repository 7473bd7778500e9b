from .spaced_sampler import SpacedSampler, space_timesteps  # noqa: F401
from .dpms_sampler import DPMSolverSampler  # noqa: F401
from .sampler import Sampler  # noqa: F401
