"""modules/scene/nerf.py:23"""
from perf_amd.distloss import eff_distloss, flatten_eff_distloss  # noqa: F401
