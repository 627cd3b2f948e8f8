import os, sys, json
sys.path.insert(0, '/root/repo')
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29597', RANK='0', WORLD_SIZE='1', PERF_DP_SINGLE_RANK='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from perf_amd import synthetic
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
torch.manual_seed(0)
scene = NeRFScene(dtype='bf16')
rays = gen_pano_rays(torch.eye(4), 512, 1024)
d_, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, d_)
seen = {'last': 0}; hits = []
def cb(kind, i):
    if i % 25 == 0 or i < 40:
        c = int(scene.sample_counters[4].item())
        if c != seen['last']:
            hits.append((kind, i, c - seen['last'])); seen['last'] = c
for ep in range(2):
    scene.train_one_episode(pool, 3000, 1500, callback=cb)
print(json.dumps({'units': scene.dp_units, 'flag_events': hits, 'total': seen['last']}))
dist.destroy_process_group()
