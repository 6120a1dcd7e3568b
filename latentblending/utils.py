from latentblending_amd.utils import *  # noqa: F401,F403
from latentblending_amd.utils import (interpolate_spherical, interpolate_linear, add_frames_linear_interp,  # noqa: F401
                                      get_spacing, get_time, compare_dicts, yml_load, yml_save)
