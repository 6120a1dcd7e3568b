from latentblending_amd.blending_engine import *  # noqa: F401,F403
from latentblending_amd.blending_engine import BlendingEngine  # noqa: F401
