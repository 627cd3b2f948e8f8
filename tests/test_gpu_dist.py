"""Data-parallel NeRFScene on real kernels without a multi-GPU node: two ranks share the one GPU (gloo carries the CUDA
gradient), and must reproduce the single-process run on the same GLOBAL batch -- SURVEY.md 8(e): every rank draws the
same index stream and keeps its slice, local losses are normalised by the global batch, ONE all-reduce of the flat
gradient (+ the sample-count slot) per step."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, out, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT)
    worker = os.path.join(ROOT, 'tests', 'dp_worker.py')
    if world == 1:
        cmd = [sys.executable, worker, out, '1024', '3']
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), worker, out, '1024', '3']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(out)


def test_two_ranks_on_one_gpu_reproduce_the_single_process_run(tmp_path):
    one = _run(1, str(tmp_path / 'w1.pt'), 0)
    two = _run(2, str(tmp_path / 'w2.pt'), 29571)
    assert one['world'] == 1 and two['world'] == 2
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))
    # the all-reduced gradient of the first step (identical parameters on both sides): 16-bit forward, fixed-point grid
    # gradient whose unit follows each rank's own max |dfeat| -> equal up to that quantisation
    assert rel(two['g_geo'], one['g_geo']) < 2e-3, rel(two['g_geo'], one['g_geo'])
    assert rel(two['g_app'], one['g_app']) < 2e-3, rel(two['g_app'], one['g_app'])
    assert one['geo_steps'] == two['geo_steps'] == 3
    # after 3 Adam steps per phase the parameters moved the same way (Adam's early steps are sign-like: an entry whose
    # tiny gradient rounds differently moves by +-lr, hence a norm test relative to the distance travelled)
    assert torch.equal(one['geo0'], two['geo0']) and torch.equal(one['app0'], two['app0'])
    for k in ('geo', 'app'):
        moved = float((one[k] - one[k + '0']).norm())
        assert moved > 0
        assert float((two[k] - one[k]).norm()) < 0.1 * moved, (k, float((two[k] - one[k]).norm()), moved)
    # a batch without samples on any rank: the optimizer step is skipped everywhere (reference: nerf.py:204-206)
    assert one['empty_batch_skipped'] and two['empty_batch_skipped']
    # the geometry step's colour render (query key 'rgb'; issued while the all-reduce is in flight under DP): rank 0 of
    # the 2-rank world holds the first half of the global batch
    per = two['first_colors'].shape[0]
    assert per * 2 == one['first_colors'].shape[0]
    assert float((two['first_colors'] - one['first_colors'][:per]).abs().max()) < 2e-3


@pytest.mark.parametrize('n_levels,log2_t', [(16, 18), (20, 20)])
def test_level_sharded_encode_matches_the_unsharded_kernel(tmp_path, n_levels, log2_t):
    """BASELINE config 5's multi-GPU split (perf_amd/sharded.py): tables cut by level over the ranks, positions
    all-gathered, features returned by one all-to-all.  Two ranks on one GPU: the features every rank gets are
    bit-identical to the unsharded encode, and the sharded table gradient equals the matching slice of the unsharded one."""
    out = str(tmp_path / 's.pt')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(29600 + n_levels), os.path.join(ROOT, 'tests', 'sharded_worker.py'), out, str(n_levels), str(log2_t)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = torch.load(out)
    assert res['world'] == 2 and all(res['fwd_equal']), res
    assert max(res['bwd_rel_err']) < 1e-5, res                       # fp32 LDS atomics: association order only
    a = res['assignment']
    assert sorted(a[0] + a[1]) == list(range(n_levels)) and abs(len(a[0]) - len(a[1])) <= 1
