"""perf_amd -- MI355X-native panoramic-NeRF render/train hot path (PeRF modules/fields + modules/scene).

Host side: thin Python over a C-ABI HIP library (include/perf_hip.h).  No CPU fallback."""
__version__ = '0.1.0'


def install_shims(scene=False):
    """Put the drop-in `tinycudann`, `nerfacc` and `torch_efficient_distloss` packages on sys.path so that
    PeRF's modules/ and core_exp_runner.py import them unchanged.

    scene=True additionally serves PeRF's own hot-path modules from this package's mirrors: `modules.scene.nerf`
    (NeRFScene), `modules.scene.nerf_renderer` (NeRFOCCRenderer, NeRFPropRenderer) and `modules.dataset.sup_info`
    (SupInfoPool) -- a sys.meta_path finder that answers for exactly these three names and lets everything else of the
    reference tree import as it is.  core_exp_runner.py:64 instantiates `globals()[conf.scene_class_name]`, i.e. whatever
    `from modules.scene.nerf import NeRFScene` bound (:24): with the finder installed an UNMODIFIED runner reaches the
    explicit kernel chains, the hipGraph-replayed steps and the one-launch occupancy build instead of the autograd
    formulation its own NeRFScene gets over the operator shims.  Same class names, same constructor
    keywords (NeRFScene(exp_dir, train_conf=, estimator_type=, renderer_conf=); SupInfoPool().register_sup_info(pose=,
    mask=, rgb=, distance=, normal=)), same methods the runner calls (fit, render, get_pano_visibility_mask,
    state_dict / load_state_dict, geo_check, gen_occ_grid)."""
    import os
    import sys
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shims')
    if d not in sys.path:
        sys.path.insert(0, d)
    if scene and not any(isinstance(f, _MirrorFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _MirrorFinder())
    return d


def uninstall_scene_shims():
    import sys
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, _MirrorFinder)]
    for name in _MirrorFinder.SERVED:
        sys.modules.pop(name, None)


class _MirrorFinder:
    """importlib finder + loader for the three reference modules perf_amd mirrors (install_shims(scene=True))."""
    SERVED = {
        'modules.scene.nerf': ('perf_amd.scene', ('NeRFScene', 'Rays', 'psnr')),
        'modules.scene.nerf_renderer': ('perf_amd.renderer', ('NeRFOCCRenderer', 'NeRFPropRenderer')),
        'modules.dataset.sup_info': ('perf_amd.scene', ('SupInfoPool',)),
    }

    def find_spec(self, fullname, path=None, target=None):
        if fullname not in self.SERVED:
            return None
        from importlib.machinery import ModuleSpec
        return ModuleSpec(fullname, self, origin='perf_amd mirror of ' + fullname)

    def create_module(self, spec):
        return None                      # default module object

    def exec_module(self, module):
        import importlib
        src_name, names = self.SERVED[module.__name__]
        src = importlib.import_module(src_name)
        for n in names:
            setattr(module, n, getattr(src, n))
        module.__perf_amd_mirror__ = src_name
