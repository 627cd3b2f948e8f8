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


def _free_port():
    import socket
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    return port


def _run(world, out, port, dp_mode='sharded', units='exact'):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT, PERF_TEST_DP_MODE=dp_mode, PERF_DP_UNITS=units)
    worker = os.path.join(ROOT, 'tests', 'dp_worker.py')
    if world == 1:
        cmd = [sys.executable, worker, out, '1024', '3']
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), worker, out, '1024', '3']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(out)


def _common_checks(one, two):
    assert one['world'] == 1 and two['world'] == 2
    assert one['geo_steps'] == two['geo_steps'] == 3
    assert torch.equal(one['geo0'], two['geo0']) and torch.equal(one['app0'], two['app0'])
    # a batch without samples on any rank: the optimizer step is skipped everywhere (reference: nerf.py:204-206)
    assert one['empty_batch_skipped'] and two['empty_batch_skipped']
    # the geometry step's colour render (query key 'rgb'; issued while the exchange is in flight under DP): rank 0 of
    # the 2-rank world holds the first half of the global batch
    per = two['first_colors'].shape[0]
    assert per * 2 == one['first_colors'].shape[0]
    assert float((two['first_colors'] - one['first_colors'][:per]).abs().max()) < 2e-3


def test_two_ranks_on_one_gpu_reproduce_the_single_process_run_bit_for_bit(tmp_path):
    """Sharded exchange (perf_amd/dp.py), EXACT units: job-wide fixed-point units + integer reduce-scatter make the summed TABLE gradient of
    two ranks equal the single-process one BIT FOR BIT (integer sums do not depend on how the samples are dealt to
    workgroups or ranks).  The 3,072 / 7,168 MLP weight gradients are fp32 sums of per-rank MFMA reductions: equal to fp32
    rounding, not to the bit."""
    one = _run(1, str(tmp_path / 'w1.pt'), 0)
    two = _run(2, str(tmp_path / 'w2.pt'), 29571)
    two_r1 = torch.load(str(tmp_path / 'w2.pt') + '.1')
    assert two['dp_mode'] == 'sharded'
    _common_checks(one, two)
    for key, n_net in (('geo', 3072), ('app', 7168)):
        g1 = one['g_' + key]
        (lo0, hi0), (lo1, hi1) = two[key + '_slice'], two_r1[key + '_slice']
        assert lo0 == 0 and hi0 == lo1 and 2 * hi1 == g1.numel() - n_net
        table = torch.cat([two['g_' + key][n_net:], two_r1['g_' + key][n_net:]])
        assert torch.equal(table, g1[n_net:]), (key, float((table - g1[n_net:]).abs().max()))
        net2, net1 = two['g_' + key][:n_net], g1[:n_net]
        assert torch.equal(net2, two_r1['g_' + key][:n_net])                       # every rank holds the same summed MLP gradient
        assert float((net2 - net1).abs().max()) <= 2e-5 * float(net1.abs().max()), key
    # after ONE step the table part of the parameters is bit-identical as well; the masters agree on both ranks
    n_net = 3072
    assert torch.equal(two['geo1'][n_net:], one['geo1'][n_net:])
    assert torch.equal(two['geo1'], two_r1['geo1']) and torch.equal(two['geo'], two_r1['geo']) and torch.equal(two['app'], two_r1['app'])
    # later steps start from MLP weights that differ in their last bits: equal to a small fraction of the distance travelled
    for k in ('geo', 'app'):
        moved = float((one[k] - one[k + '0']).norm())
        assert moved > 0
        assert float((two[k] - one[k]).norm()) < 0.02 * moved, (k, float((two[k] - one[k]).norm()), moved)
    assert two['counters'][4] == 0 and two['counters'][5] == 0                  # nothing overflowed, nothing was truncated


def test_eight_ranks_on_one_gpu_sharded_exchange(tmp_path):
    """The world size the scaling run uses: EIGHT ranks (sharing this box's one GPU over gloo) through the sharded exchange with
    exact units -- slices of 415,076 table entries per rank, 128 rays per rank of the 1,024-ray global batch -- reproduce the
    single process's table gradient bit for bit, hold identical parameters on every rank after three steps of each network, and
    skip the all-empty step alike."""
    one = _run(1, str(tmp_path / 'w1.pt'), 0)
    eight = _run(8, str(tmp_path / 'w8.pt'), _free_port())
    others = [torch.load(str(tmp_path / 'w8.pt') + f'.{r}') for r in range(1, 8)]
    assert eight['world'] == 8 and eight['geo_steps'] == one['geo_steps'] == 3 and eight['empty_batch_skipped']
    for key, n_net in (('geo', 3072), ('app', 7168)):
        ranks = [eight] + others
        bounds = [r[key + '_slice'] for r in ranks]
        assert bounds[0][0] == 0 and all(bounds[i][1] == bounds[i + 1][0] for i in range(7)) and 2 * bounds[7][1] == one['g_' + key].numel() - n_net
        table, ref = torch.cat([r['g_' + key][n_net:] for r in ranks]), one['g_' + key][n_net:]
        if key == 'geo':                  # the first geometry step starts from identical parameters: integer sums, to the bit
            assert torch.equal(table, ref)
        else:                             # the colour field's first step follows three geometry steps (fp32-rounding apart)
            assert float((table - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    for r in others:
        assert torch.equal(r['geo'], eight['geo']) and torch.equal(r['app'], eight['app'])
    for k in ('geo', 'app'):
        moved = float((one[k] - one[k + '0']).norm())
        assert float((eight[k] - one[k]).norm()) < 0.02 * moved, (k, float((eight[k] - one[k]).norm()), moved)
    assert eight['counters'][4] == 0 and eight['counters'][5] == 0


def test_two_ranks_with_lagged_units(tmp_path):
    """The default exchange: the units of step t come from the statistics of step t-1 (no collective between the MLP backward
    and the grid backward), one bit coarser.  The first step has nothing to lag behind and takes the exact path -- its summed
    table gradient equals the single process's bit for bit; later steps quantise the same gradient with a coarser unit: the
    parameters stay within a small fraction of the distance travelled, both ranks hold the same ones, nothing is skipped."""
    one = _run(1, str(tmp_path / 'w1.pt'), 0)
    two = _run(2, str(tmp_path / 'w2.pt'), 29575, units='lagged')
    two_r1 = torch.load(str(tmp_path / 'w2.pt') + '.1')
    _common_checks(one, two)
    table = torch.cat([two['g_geo'][3072:], two_r1['g_geo'][3072:]])
    assert torch.equal(table, one['g_geo'][3072:])                                  # the very first step: exact path
    # (the colour network's first gradient already sees a density field trained with two lagged steps: equal to that)
    table = torch.cat([two['g_app'][7168:], two_r1['g_app'][7168:]])
    assert float((table - one['g_app'][7168:]).norm()) < 2e-2 * float(one['g_app'][7168:].norm())
    assert torch.equal(two['geo'], two_r1['geo']) and torch.equal(two['app'], two_r1['app'])
    for k in ('geo', 'app'):
        moved = float((one[k] - one[k + '0']).norm())
        assert float((two[k] - one[k]).norm()) < 0.02 * moved, (k, float((two[k] - one[k]).norm()), moved)
    assert two['counters'][4] == 0 and two['counters'][5] == 0


def test_two_ranks_plain_allreduce_mode(tmp_path):
    """dp_mode = 'allreduce': one all-reduce of the flat fp32 gradient (+ the sample-count slot), Adam on every rank; each
    rank's fixed-point unit follows its own max |dfeat| -> equal to that quantisation."""
    one = _run(1, str(tmp_path / 'w1.pt'), 0, 'allreduce')
    two = _run(2, str(tmp_path / 'w2.pt'), 29573, 'allreduce')
    assert two['dp_mode'] == 'allreduce'
    _common_checks(one, two)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))
    assert rel(two['g_geo'], one['g_geo']) < 2e-3, rel(two['g_geo'], one['g_geo'])
    assert rel(two['g_app'], one['g_app']) < 2e-3, rel(two['g_app'], one['g_app'])
    for k in ('geo', 'app'):
        moved = float((one[k] - one[k + '0']).norm())
        assert float((two[k] - one[k]).norm()) < 0.1 * moved, (k, float((two[k] - one[k]).norm()), moved)


@pytest.mark.parametrize('units', ['exact', 'lagged'])
def test_rccl_exchange_on_a_world_of_one(tmp_path, units):
    """The real RCCL backend on this box's one GPU: a world of ONE rank takes the sharded data-parallel path
    (PERF_DP_SINGLE_RANK=1) -- [statistics all-gather,] int32 reduce-scatter, small all-reduce with the job-wide gate, Adam on
    the slice, all-gather of the 16-bit copy, captured with the step in one hipGraph when the capture probe passes.  Exact
    units: two consecutive episodes end with exactly the parameters of the plain single process (the table gradient is an
    integer sum either way; with one rank the MLP gradient is the same fp32 sum too).  Lagged units (the default): the same
    gradient quantised with a one-bit coarser unit -- parameters within a small fraction of the distance travelled."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT, PERF_DP_UNITS=units)
    worker = os.path.join(ROOT, 'tests', 'rccl_single_worker.py')
    res = {}
    for mode in ('plain', 'rccl'):
        out = str(tmp_path / f'{mode}.pt')
        r = subprocess.run([sys.executable, worker, out, mode, str(_free_port())], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        res[mode] = torch.load(out)
    assert not res['plain']['dist'] and res['rccl']['dist']
    print('RCCL collectives captured in the step graph:', res['rccl']['graph_verdict'])
    assert res['rccl']['graph_verdict'] is not None                       # the probe ran (its verdict decides graph vs eager)
    assert res['plain']['rng_counter'] == res['rccl']['rng_counter'] == 23 + 17
    if units == 'exact':
        assert torch.equal(res['plain']['geo'], res['rccl']['geo'])
        assert torch.equal(res['plain']['app'], res['rccl']['app'])
    else:
        for k in ('geo', 'app'):
            a, b = res['plain'][k].float(), res['rccl'][k].float()
            assert float((a - b).norm()) < 0.05 * float(a.norm()), (k, float((a - b).norm()), float(a.norm()))
    assert res['rccl']['counters'][4] == 0 and res['rccl']['counters'][5] == 0


@pytest.mark.parametrize('n_ranks', [2, 8])
def test_bench_launches_its_own_ranks(tmp_path, n_ranks):
    """`python bench.py --gpus N` without a launcher (what the driver's scaling run does): bench.py spawns the ranks itself.
    N = 2 and N = 8 (the node the scaling run uses) ranks share this box's one GPU over gloo (PERF_BENCH_ONE_DEVICE /
    PERF_BENCH_BACKEND): a dry run of the whole N > 1 line -- argument plumbing, the eager headline, the strong-scaling block of
    BASELINE config 3, `comm`, the data-parallel PSNR episode -- so that the first contact with an 8-GPU node cannot die in
    anything but RCCL itself."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT, PERF_BENCH_ONE_DEVICE='1', PERF_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n_ranks), '--steps', '3', '--warmup', '1', '--rays-per-gpu', '1024',
           '--sustain-seconds', '0', '--psnr-geo-iters', '40', '--psnr-app-iters', '30', '--height', '128', '--width', '256']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == n_ranks and line['scaling'] == 'weak' and line['value'] > 0
    assert line['metric'].startswith('ray-samples/sec') and line['unit'] == 'ray-samples/s' and line['higher_is_better'] is True
    assert line['steps'] == 3 and line['warmup'] == 1 and line['ms_per_step'] > 0 and line['data'] == 'synthetic'
    assert line['config']['rays_per_gpu_per_step'] == 1024 and line['config']['launch'] == 'eager'      # (gloo: no graph capture)
    assert f'dp{n_ranks}' in line['config']['parallelism'] and abs(line['config']['per_gpu_value'] * n_ranks - line['value']) < 1e-6 * line['value']
    assert line['config']['kept_samples_per_gpu_per_step'] > 0
    assert line['strong']['scaling'] == 'strong' and line['strong']['global_batch_rays'] == 1024 and line['strong']['value'] > 0
    assert line['strong']['rays_per_gpu_per_step'] == 1024 // n_ranks
    assert line['psnr'] is not None and line['health'] == {'skipped_for_overflow': 0, 'skipped_for_truncation': 0}
    assert set(line['psnr']['curve']) == {'app_iter_0', 'app_iter_10', 'app_iter_30'} and line['psnr']['launch'] == 'eager'
    assert not any(' failed (' in n for n in line.get('notes', [])), line.get('notes')          # no optional measurement fell over
    assert line['roofline'] is not None and line['roofline']['kernel'] in ('perf_hashgrid_fwd', 'perf_hashgrid_bwd') and 0 < line['roofline']['frac'] < 1
    # the `comm` block a reader attributes a missed scaling target with: per-collective times of the sharded exchange, the plain
    # single-GPU step on the same per-rank workload, the difference, and the single all-reduce exchange for comparison
    comm = line['comm']
    assert set(comm['ms_per_step_per_collective']) == {'reduce_scatter', 'all_reduce_small', 'all_gather_w16'} and comm['bytes']['units'] == 'lagged'
    assert comm['single_rank_step_ms'] > 0 and abs(comm['exposed_comm_ms'] - (line['ms_per_step'] - comm['single_rank_step_ms'])) < 1e-9
    assert comm['other_exchange']['dp_mode'] == 'allreduce' and comm['other_exchange']['value'] > 0
    # ... with ITS collective's time and the bf16-payload variant beside it: the first run on real links gives an A/B, not one number
    assert comm['other_exchange']['ms_per_step_per_collective']['all_reduce_flat_gradient'] > 0
    assert comm['other_exchange']['bf16_payload']['value'] > 0 and comm['other_exchange']['bf16_payload']['payload_bytes'] * 2 == comm['other_exchange']['payload_bytes']
    assert line['faithful'] is None or 'geo_ms_per_step' in line['faithful']


def test_data_parallel_soak_drops_no_step(tmp_path):
    """The lagged-units default of the sharded exchange over SEVERAL episodes (reset_geo, a fresh occupancy, new optimizers, re-captured
    capacities every time): two ranks on this box's one GPU over gloo, five episodes of the schedule on which round 5's PSNR test
    caught the job-wide gate dropping 10-12 of 300 geometry steps (256x512 doorway panorama, 1024-ray batches, 300 + 300 iterations:
    late batches whose depth loss vanishes) -- no step may be dropped for an overflow or a truncation in any episode, every rank must
    leave every episode with the same parameters, and the scene must train every time (tools/soak_episodes.py --one-device)."""
    import json
    out = str(tmp_path / 'dp_soak.json')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT)
    env.pop('PERF_DP_UNITS', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tools', 'soak_episodes.py'), '--one-device', '--episodes', '5', '--scene', 'doorway',
           '--geo', '300', '--app', '300', '--height', '256', '--width', '512', '--batch', '1024', '--out', out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert res['world'] == 2 and res['data_parallel'] and res['dp_units'] == 'lagged' and len(res['episodes']) == 5
    print('[dp soak]', [(e['psnr_dB'], e['skipped_for_overflow'], e['skipped_for_truncation'], e['seconds']) for e in res['episodes']])
    for e in res['episodes']:
        assert e['skipped_for_overflow'] == 0 and e['skipped_for_truncation'] == 0, e
        assert e['ranks_hold_identical_parameters'], e
        assert e['psnr_dB'] > 22.0, e
    # (the colour field is NOT reset between episodes -- nerf.py:170 re-instantiates the density field only -- so PSNR keeps rising)
    assert res['episodes'][-1]['psnr_dB'] >= res['episodes'][0]['psnr_dB']


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


@pytest.mark.parametrize('world,log2_t,rows,layout', [(2, 20, 2, 'tcnn'), (4, 22, 1, 'tcnn'), (2, 21, 2, 'line_local'), (2, 21, 2, 'line_overlap')])
def test_config5_row_shard_through_the_level_sharded_path(tmp_path, world, log2_t, rows, layout):
    """BASELINE config 5's inference batch at its stated shape -- rows of a 4096x2048 panorama, 256 samples per ray, L = 20
    hash grids (T = 2^20 / 2^22 here: the box is shared by all ranks AND the unsharded comparison copies) -- rendered by two
    and by FOUR ranks through the level-sharded fields (perf_amd/sharded.py:LevelShardedNeRF: greedy level split of a
    non-trivial world, all-to-all blocks of unequal row counts): pixels, distances and opacities are bit-identical to the
    unsharded render of the same rays, on every rank."""
    out = str(tmp_path / 'c5.pt')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'config5_worker.py'), out, str(rows), str(log2_t), layout]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = torch.load(out)
    assert len(res) == world
    for rr in res:
        assert rr['rays'] == rows * 4096 and rr['marched'] == rows * 4096 * 256
        assert 0 < rr['kept'] <= rr['marched'] and all(rr['same'].values()), rr
        assert rr['rgb_range'][1] > rr['rgb_range'][0]
    held = sorted(l for rr in res for l in rr['levels'])
    assert held == list(range(20))
    assert max(len(rr['levels']) for rr in res) - min(len(rr['levels']) for rr in res) <= 1
