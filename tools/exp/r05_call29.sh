R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python - <<'PY' 2>&1 | tail -60
import cProfile, pstats, io, sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from perf_amd import synthetic
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
torch.manual_seed(0)
scene = NeRFScene(dtype='bf16')
H, W = 1024, 2048
rays = gen_pano_rays(torch.eye(4), H, W)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
scene.train_one_episode(pool, 300, 150)       # warm everything
torch.cuda.synchronize()
marks = []
def cb(phase, i):
    if i in (0, 2, 3, 4, 70, 200, 2999) or (phase == 'app' and i in (0, 2, 3, 4, 1499)):
        torch.cuda.synchronize(); marks.append((phase, i, time.perf_counter()))
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
pr.enable()
scene.train_one_episode(pool, 3000, 1500, callback=cb)
torch.cuda.synchronize()
pr.disable()
t1 = time.perf_counter()
print('episode', t1 - t0)
prev = t0
for ph, i, t in marks:
    print(ph, i, round((t - prev) * 1e3, 2), 'ms since previous mark'); prev = t
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumtime').print_stats(28); print(s.getvalue()[-5000:])
PY
