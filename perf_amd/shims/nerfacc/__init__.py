"""`from nerfacc import ...` resolves here (modules/scene/nerf_renderer.py:5)."""
from perf_amd.nerfacc_impl import (accumulate_along_rays, render_transmittance_from_alpha,  # noqa: F401
                                   render_weight_from_density)
from . import estimators  # noqa: F401

__version__ = '0.5.3+perf_amd'
