"""Container-only developer check (needs /root/reference; never runs on the GPU box): the reference's own
`modules/fields/ngp_nerf.py` and `modules/scene/nerf_renderer.py` import and construct over this repo's shim
packages (tinycudann / nerfacc / torch_efficient_distloss), i.e. the constructor signatures and keyword names match.
Packages the image lacks are replaced by inert stubs for their *import* only.  Without a GPU the tcnn module stops at
parameter allocation with the loud "needs a HIP device" error -- that is the expected outcome here."""
import importlib, os, sys, types
os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import perf_amd

perf_amd.install_shims()
for name in ('icecream', 'kornia', 'cv2', 'trimesh', 'trimesh.creation', 'imageio', 'tensorboard', 'omegaconf', 'hydra'):
    sys.modules[name] = types.ModuleType(name)
sys.modules['icecream'].ic = lambda *a, **k: None
sys.path.insert(0, '/root/reference')
import tinycudann, nerfacc                                              # noqa: E402  (resolve to perf_amd/shims)
assert 'perf_amd/shims' in tinycudann.__file__ and 'perf_amd/shims' in nerfacc.__file__
aabb = torch.tensor([-1., -1, -1, 1, 1, 1])
rend = importlib.import_module('modules.scene.nerf_renderer')
print('NeRFOCCRenderer over the shims:', rend.NeRFOCCRenderer(1.0, 'rand_noise').bg_color)
from nerfacc.estimators.occ_grid import OccGridEstimator               # noqa: E402
print('OccGridEstimator state_dict keys:', list(OccGridEstimator(aabb, resolution=256, levels=1).state_dict().keys()))
ngp = importlib.import_module('modules.fields.ngp_nerf')
try:
    net = ngp.NGPNeRF(aabb=aabb)
    print('NGPNeRF over the shims:', [(k, tuple(v.shape)) for k, v in net.state_dict().items()])
except RuntimeError as e:
    print('NGPNeRF reached parameter allocation of the shim tcnn module:', e)
