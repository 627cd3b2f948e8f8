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
sys.modules['trimesh.creation'].icosphere = None      # utils/camera_utils.py:3 (unused by ray generation)
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

# ---- install_shims(scene=True): the three hot-path modules of the reference are served from the mirrors, everything else of
#      the reference tree imports as it is; the mirrors accept the runner's constructor keywords with the reference's own config
import inspect
import yaml
for k in [k for k in sys.modules if k.startswith('modules.scene') or k.startswith('modules.dataset')]:
    del sys.modules[k]
perf_amd.install_shims(scene=True)
nerf_mod = importlib.import_module('modules.scene.nerf')             # what core_exp_runner.py:24 imports
sup_mod = importlib.import_module('modules.dataset.sup_info')         # :20
rend_mod = importlib.import_module('modules.scene.nerf_renderer')
assert nerf_mod.__perf_amd_mirror__ == 'perf_amd.scene' and sup_mod.__perf_amd_mirror__ == 'perf_amd.scene'
assert nerf_mod.NeRFScene.__module__ == 'perf_amd.scene' and rend_mod.NeRFOCCRenderer.__module__ == 'perf_amd.renderer'
iface = importlib.import_module('modules.scene.scene')               # the abstract interface: still the reference's own file
assert iface.__file__.startswith('/root/reference/'), iface.__file__
conf = yaml.safe_load(open('/root/reference/configs/nerf.yaml'))
assert conf['scene_class_name'] == 'NeRFScene'
bound = inspect.signature(nerf_mod.NeRFScene.__init__).bind(None, '/tmp/exp', **conf['scene'])      # core_exp_runner.py:64
print('NeRFScene(exp_dir, **conf.scene) binds on the mirror:', sorted(bound.arguments)[:6])
need = {'fit', 'render', 'get_pano_visibility_mask', 'state_dict', 'load_state_dict', 'set_train', 'set_eval'}
assert need <= set(dir(nerf_mod.NeRFScene)), need - set(dir(nerf_mod.NeRFScene))
inspect.signature(sup_mod.SupInfoPool.register_sup_info).bind(None, pose=0, mask=0, rgb=0, distance=0, normal=0)   # :77-82
assert {'gen_occ_grid', 'geo_check', 'state_dict', 'load_state_dict'} <= set(dir(sup_mod.SupInfoPool))
used = set()
import re
src = open('/root/reference/core_exp_runner.py').read()
for m in re.finditer(r'self\.scene\.(\w+)', src):
    used.add(m.group(1))
missing = {u for u in used if not hasattr(nerf_mod.NeRFScene, u)}
print('scene methods core_exp_runner.py calls:', sorted(used), '-> missing on the mirror:', sorted(missing))
assert not missing
used = {m.group(1) for m in re.finditer(r'self\.sup_pool\.(\w+)', src)}
missing = {u for u in used if not hasattr(sup_mod.SupInfoPool, u)}
print('pool methods core_exp_runner.py calls:', sorted(used), '-> missing on the mirror:', sorted(missing))
assert not missing
print('install_shims(scene=True): OK')
