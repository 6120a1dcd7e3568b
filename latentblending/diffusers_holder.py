from latentblending_amd.diffusers_holder import *  # noqa: F401,F403
from latentblending_amd.diffusers_holder import DiffusersHolder  # noqa: F401
