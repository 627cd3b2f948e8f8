"""perf_amd -- MI355X-native panoramic-NeRF render/train hot path (PeRF modules/fields + modules/scene).

Host side: thin Python over a C-ABI HIP library (include/perf_hip.h).  No CPU fallback."""
__version__ = '0.1.0'


def install_shims():
    """Put the drop-in `tinycudann`, `nerfacc` and `torch_efficient_distloss` packages on sys.path so that
    PeRF's modules/ and core_exp_runner.py import them unchanged."""
    import os
    import sys
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shims')
    if d not in sys.path:
        sys.path.insert(0, d)
    return d
