"""perf_amd -- MI355X-native panoramic-NeRF render/train hot path (PeRF modules/fields + modules/scene).

Host side: thin Python over a C-ABI HIP library (include/perf_hip.h).  No CPU fallback."""
__version__ = '0.1.0'
