"""modules/scene/nerf_renderer.py:6, modules/scene/nerf.py:24"""
from perf_amd.nerfacc_impl import PropNetEstimator  # noqa: F401
