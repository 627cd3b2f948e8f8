"""modules/scene/nerf_renderer.py:7, modules/scene/nerf.py:25"""
from perf_amd.nerfacc_impl import OccGridEstimator  # noqa: F401
