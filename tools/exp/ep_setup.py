import sys, time, torch
sys.path.insert(0, '/root/repo')
from perf_amd import synthetic
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
scene = NeRFScene(dtype='bf16')
rays = gen_pano_rays(torch.eye(4), 1024, 2048)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
def T(f, name):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); print(name, round((time.perf_counter() - t) * 1e3, 1), 'ms'); return r
for rep in range(2):
    scene.set_train()
    T(lambda: scene.prepare_occupancy(pool), 'prepare_occupancy')
    T(lambda: pool.gen_occ_grid(256), '  of which gen_occ_grid')
    T(lambda: scene.nerf.reset_geo(), 'reset_geo')
    T(lambda: scene.make_optimizer(scene.nerf.geo_mlp, 0.0), 'make_optimizer')
