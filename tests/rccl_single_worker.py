"""Worker of tests/test_gpu_dist.py::test_rccl_exchange_on_a_world_of_one: a short training episode either as a plain single
process or as a data-parallel world of ONE rank on the real RCCL backend (PERF_DP_SINGLE_RANK=1) -- reduce-scatter of the
int32 gradient fields, sharded Adam, all-gather of the 16-bit copy, all captured in the step's hipGraph when the probe
allows.  Writes the resulting parameters.   python tests/rccl_single_worker.py <out.pt> plain|rccl <port>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    out_path, mode, port = sys.argv[1], sys.argv[2], sys.argv[3]
    torch.cuda.set_device(0)
    if mode == 'rccl':
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK='0', WORLD_SIZE='1', PERF_DP_SINGLE_RANK='1')
        os.environ.setdefault('TORCH_NCCL_RETHROW_CUDA_ERRORS', '0')       # (as bench.py: see the note there)
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    from perf_amd import synthetic, scene as S
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
    torch.manual_seed(0)
    scene = NeRFScene(dtype='bf16')
    rays = gen_pano_rays(torch.eye(4), 64, 128)
    d_, rgb = synthetic.room(rays.d)
    pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, d_)
    scene.train_conf.pixel_loss_batch_size = 1024
    scene.train_one_episode(pool, 14, 9)
    scene.train_one_episode(pool, 10, 7)         # a second episode: new geometry field, new optimizers, the colour table's state carries over
    torch.cuda.synchronize()
    info = {'geo': scene.nerf.geo_mlp.params.detach().cpu(), 'app': scene.nerf.app_mlp.params.detach().cpu(), 'mode': mode,
            'dist': scene._dist()[0] is not None, 'graph_verdict': S._DP_GRAPH_VERDICT, 'counters': scene.sample_counters.tolist(),
            'rng_counter': int(scene._rng_counter.item()) if scene._rng_counter is not None else None}
    torch.save(info, out_path)
    if mode == 'rccl':
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
