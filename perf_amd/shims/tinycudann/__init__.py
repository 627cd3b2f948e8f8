"""`import tinycudann as tcnn` resolves here when perf_amd/shims is on sys.path (perf_amd.install_shims()).
PeRF call sites: modules/fields/ngp_nerf.py:14,96,116,179,230; modules/geo_predictors/pano_joint_predictor.py:13,30;
modules/geo_predictors/pano_geo_refiner.py:6,19."""
from perf_amd.tcnn import Encoding, NetworkWithInputEncoding  # noqa: F401

__version__ = '1.7+perf_amd'
