"""CPU: the PSNR bounds of tests/psnr_bounds.py against the committed evidence -- the oracle fixtures carry what the bounds are
computed from, and the HIP ensembles recorded on the GPU (profiles/r06_psnr_ensemble.json: `python tools/psnr_parity.py ensemble`)
meet them through the same checker the GPU tests call on fresh runs."""
import json
import os

import pytest

from tests import psnr_bounds as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = {'room': 'psnr_curve.json', 'doorway': 'psnr_curve_doorway.json', 'pillars': 'psnr_curve_pillars.json'}


def _golden(name):
    return json.load(open(os.path.join(ROOT, 'tests', 'golden', FAMILIES[name])))


@pytest.mark.parametrize('family', sorted(FAMILIES))
def test_the_fixtures_carry_what_the_bounds_are_made_of(family):
    g = _golden(family)
    sp = g['oracle_spread']['max_abs_delta_db']
    for k in B.marks_of(g):
        assert 0.0 < sp[k] < 0.25                                                  # the fp32 oracle under a one-ulp change of its initialisation
    emu = g['oracle_16bit']['bf16']['curves']
    assert sorted(emu) == sorted(str(r['seed']) for r in g['seeds'])               # the bf16-storage oracle, every seed (15-45 min per seed on the CPU)
    for dtype in ('bf16', 'fp16'):
        for k, b in B.seed_bound(g, dtype).items():
            assert b['base'] == max(0.1, sp[k])
            assert b['bound'] == b['base'] + b['storage_noise']
            # what 8 mantissa bits of storage do to the oracle is of the size of the north_star's tolerance, not beyond it; no
            # emulated curve is committed for fp16, the reference's own type: its bound is the base
            assert (0.005 < b['storage_noise'] < 0.15) if dtype == 'bf16' else b['storage_noise'] == 0.0


@pytest.mark.parametrize('family', sorted(FAMILIES))
def test_the_recorded_hip_ensembles_meet_the_bounds(family):
    g = _golden(family)
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'r06_psnr_ensemble.json')))['families'][family]
    oracle = {r['seed']: r['oracle'] for r in g['seeds']}
    for dtype in ('bf16', 'fp16'):
        by_seed = {}
        for row in rec[dtype]:
            sd = row['seed']
            by_seed[sd] = [{k: oracle[sd][k] + row['hip_minus_oracle'][m][k] for k in B.marks_of(g)} for m in ('nominal', 'one_ulp_up', 'one_ulp_down')]
        assert sorted(by_seed) == sorted(oracle)
        out = B.check_family(g, dtype, by_seed)
        if dtype == 'fp16':                                                        # the reference's storage type: max(0.1 dB, oracle_spread) per seed, no allowance
            for k, b in B.seed_bound(g, dtype).items():
                assert max(abs(v) for v in out[k]['ensemble_mean']) <= b['base']


def test_a_broken_bound_is_reported():
    g = _golden('room')
    by_seed = {r['seed']: [dict(r['oracle'])] for r in g['seeds']}
    B.check_family(g, 'bf16', by_seed, log=lambda *_: None)                        # the oracle itself: zero deltas
    sd = g['seeds'][0]['seed']
    by_seed[sd] = [{k: v + (0.3 if k.startswith('psnr') else 0.0) for k, v in g['seeds'][0]['oracle'].items()}]
    with pytest.raises(AssertionError):
        B.check_family(g, 'bf16', by_seed, log=lambda *_: None)
