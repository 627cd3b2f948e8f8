"""Worker of tests/test_gpu_dist.py: `steps` data-parallel geometry + colour steps of NeRFScene on a world of WORLD_SIZE
ranks that SHARE one GPU (gloo moves the CUDA gradient through the host: the test is about the sharding, normalisation
and skip logic of perf_amd/scene.py, not about RCCL).  Rank 0 writes the resulting parameters and the first all-reduced
gradient.   python -m torch.distributed.run --nproc-per-node 2 ... tests/dp_worker.py <out.pt> <global_batch> <steps>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_path, global_batch, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group('gloo')
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
    torch.manual_seed(0)
    scene = NeRFScene(dtype='fp16')
    rays = gen_pano_rays(torch.eye(4), 64, 128)
    d_, rgb = synthetic.room(rays.d)
    pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, d_)
    scene.train_conf.pixel_loss_batch_size = global_batch
    scene.set_train(); scene.prepare_occupancy(pool); scene.nerf.reset_geo()
    init = {'geo0': scene.nerf.geo_mlp.params.detach().cpu().clone(), 'app0': scene.nerf.app_mlp.params.detach().cpu().clone()}
    gen = torch.Generator(device='cuda'); gen.manual_seed(77)                 # the same index stream on every rank
    cg = torch.Generator().manual_seed(5)
    full = {'jitter': torch.rand(global_batch, generator=cg), 'noise': torch.rand(global_batch, 1, generator=cg),
            'bg': torch.rand(global_batch, 3, generator=cg)}
    per = global_batch // world
    rand = {k: v[rank * per:(rank + 1) * per].cuda().contiguous() for k, v in full.items()}
    first_grad = {}
    orig = scene._apply_grad
    scene.dp_mode = os.environ.get('PERF_TEST_DP_MODE', 'sharded')

    def spy(net, grad, optimizer, dist_info, overlap, **kw):
        orig(net, grad, optimizer, dist_info, overlap, **kw)
        first_grad.setdefault('geo' if net is scene.nerf.geo_mlp else 'app', grad[:net.params.numel()].detach().clone())
    scene._apply_grad = spy
    # sharded exchange: the summed gradient only ever exists in slices -- record [MLP part | this rank's slice] after the
    # exchange of the first step of each network (tests/test_gpu_dist.py puts the slices together)
    from perf_amd import dp as _dp
    orig_rs = _dp.ShardedExchange.reduce_and_step

    def spy_rs(self, dw, opt, **kw):
        out = orig_rs(self, dw, opt, **kw)
        key = 'geo' if opt.net is scene.nerf.geo_mlp else 'app'
        n_own = 2 * (self.hi - self.lo)
        first_grad.setdefault(key, torch.cat([self.dw[:self.n_net], self.shard.view(torch.float32)[:n_own]]).detach().clone())
        first_grad.setdefault(key + '_slice', (self.lo, self.hi))
        return out
    _dp.ShardedExchange.reduce_and_step = spy_rs
    opt = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
    for i in range(steps):
        scene.update_lr(opt, scene.train_conf.geo_optimizer, 0.1)
        scene.train_one_step_geo(opt, pool, progress=0.5, rand=rand, generator=gen, prefetch_next=i + 1 < steps)
        if i == 0:
            first_colors = scene.last_colors.detach().float().cpu().clone()      # the step's colour render (deferred under DP)
            scene.sync_params()
            geo1 = scene.nerf.geo_mlp.params.detach().cpu().clone()              # parameters after ONE step
    opt2 = scene.make_optimizer(scene.nerf.app_mlp, 0.0)
    for i in range(steps):
        scene.update_lr(opt2, scene.train_conf.app_optimizer, 0.1)
        scene.train_one_step_app(opt2, pool, progress=0.5, rand=rand, generator=gen)
    # a step whose batch has no samples anywhere must be skipped on every rank (step counts stay)
    scene.estimator.set_binaries(torch.zeros(256 ** 3, dtype=torch.uint8, device='cuda'))
    before = scene.nerf.geo_mlp.params.detach().clone()
    opt3 = scene.make_optimizer(scene.nerf.geo_mlp, 1e-2)
    scene._geo_pre = None
    scene.train_one_step_geo(opt3, pool, progress=0.5, rand=rand, generator=gen, prefetch_next=False)
    torch.cuda.synchronize()
    skipped = bool(torch.equal(before, scene.nerf.geo_mlp.params.detach())) and opt3.step_count == 0
    scene.sync_params()
    torch.save({'geo': scene.nerf.geo_mlp.params.detach().cpu(), 'app': scene.nerf.app_mlp.params.detach().cpu(), 'geo1': geo1,
                'g_geo': first_grad['geo'].cpu(), 'g_app': first_grad['app'].cpu(), 'geo_slice': first_grad.get('geo_slice'),
                'app_slice': first_grad.get('app_slice'), 'empty_batch_skipped': skipped, 'first_colors': first_colors,
                'geo_steps': opt.step_count, 'world': world, 'dp_mode': scene.dp_mode, 'counters': scene.sample_counters.tolist(), **init},
               out_path if rank == 0 else out_path + f'.{rank}')
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
