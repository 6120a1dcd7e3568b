"""Import-compatibility shim: scripts written for lunarring/latentblending
(``from latentblending.blending_engine import BlendingEngine``) resolve to the MI355X-native
implementation in ``latentblending_amd``."""
from latentblending_amd import *  # noqa: F401,F403
from latentblending_amd import BlendingEngine, DiffusersHolder  # noqa: F401
