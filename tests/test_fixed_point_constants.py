"""The constants of the fixed-point grid gradient, DERIVED instead of remembered (CPU; no GPU, no library call).

perf_amd/csrc/grid_fixed_point.hpp keeps the largest field of a level inside [2^21, 2^25) units, hashgrid_bwd.hip raises the overflow
flag at 2^29; each number was set after a failure on the GPU.  This file restates the closed loop in Python -- with
the constants PARSED from the sources, so the test follows them -- and checks the properties the numbers have to have together:

  1. integer fields are sums modulo 2^32: a partial sum (one replica, one rank, one arrival order) may wrap as long as the FINAL sum
     fits -- which is why the flag is only ever tested on final sums, and why 2^29 (not 2^31) is enough of a limit: the flag level
     plus the largest single contribution still fits an int32;
  2. the band is wider than the loop's granularity: from any starting headroom a stationary gradient is brought into the band and
     STAYS there (no oscillation: a correction never overshoots the other edge), upward in one call, downward one bit per call;
  3. the margin between the band and the flag: a level whose largest sum grows by less than 2^(29 - 25) = 16x from one call to the
     next is never flagged (the soak that motivated 16x saw 4x jumps flag 8 of 112,500 steps with a 4x margin);
  4. the resolution that is left: inside the band the largest contribution of a call is at least 2^(31 - 28) = 8 units, and with the
     static fan-in guess (no state) never less than 2^7 units.
"""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, '..', 'perf_amd', 'csrc')


def _const(text, name):
    m = re.search(r'\b' + name + r'\s*=\s*(-?\d+)', text)
    assert m, name
    return int(m.group(1))


def _constants():
    hpp = open(os.path.join(CSRC, 'grid_fixed_point.hpp')).read()
    bwd = open(os.path.join(CSRC, 'hashgrid_bwd.hip')).read()
    flags = set(re.findall(r'field_max >= \(1 << (\d+)\)', bwd))
    assert len(flags) == 1, flags                      # every owner path and the replica reduction flag at the same level
    clamp = re.search(r'h = h < (\d+) \? \1 : \(h > (\d+) \? \2 : h\);\s*\} else \{\s*h = h < (\d+) \? \3 : \(h > (\d+) \? \4 : h\);', hpp)
    assert clamp, 'fixed_point_shift: the clamps of the headroom moved'
    floor = re.search(r'adj > (-\d+)\) adj -= 1', hpp)
    assert floor
    return dict(top=_const(hpp, 'kHeadroomTopBit'), low=_const(hpp, 'kHeadroomLowBit'), bias=_const(hpp, 'kHeadroomStartBias'),
                flag=int(flags.pop()), h_min=int(clamp.group(1)), h_max=int(clamp.group(2)), h_min_static=int(clamp.group(3)),
                h_max_static=int(clamp.group(4)), adj_floor=int(floor.group(1)), max_replicas=_const(bwd, 'kMaxReplicas'))


C = _constants()


def headroom_feedback(adj, fm):
    """grid_fixed_point.hpp:headroom_feedback"""
    if fm >= (1 << C['top']):
        adj += fm.bit_length() - C['top'] + 1
    elif fm < (1 << C['low']) and adj > C['adj_floor']:
        adj -= 1
    return adj


def headroom_bits(fan, adj=None):
    """grid_fixed_point.hpp:fixed_point_shift, the headroom h (the unit of a level is 2^(e - 31 + h) for max |dfeat| < 2^e)."""
    h = (0 if fan <= 1 else (fan - 1).bit_length()) + 6
    if adj is None:
        return min(max(h, C['h_min_static']), C['h_max_static'])
    return min(max(h + adj + C['bias'], C['h_min']), C['h_max'])


def test_partial_sums_may_wrap_only_the_final_sum_has_to_fit():
    rng = np.random.default_rng(3)
    big = (1 << 31) - 1
    for _ in range(200):
        n = int(rng.integers(2, 4000))
        contrib = rng.integers(-(1 << 28), 1 << 28, size=n, dtype=np.int64)
        contrib -= contrib.sum() // n                                  # a final sum of modest size from partials that do not stay modest
        true = int(contrib.sum())
        assert abs(true) <= big
        for parts in (1, 2, C['max_replicas']):                        # replicas / ranks: any partition, any order inside a part
            order = rng.permutation(n)
            slabs = [np.int32(0)] * parts
            with np.errstate(over='ignore'):
                for k, i in enumerate(order):
                    slabs[k % parts] = np.int32(slabs[k % parts] + np.int32(contrib[i]))      # wraps like the LDS / global integer adds
                total = np.int32(0)
                for s in slabs:
                    total = np.int32(total + s)
            assert int(total) == true
    # the flag level leaves room for one more contribution of ANY size the unit allows (a contribution is < 2^(31 - h_min) units)
    assert (1 << C['flag']) + (1 << (31 - C['h_min'])) <= big + 1
    # ... and a sum that was NOT flagged is exact: below the flag level the int32 never wrapped in the final value
    assert (1 << C['flag']) < big


def test_the_loop_reaches_the_band_and_stays_there():
    top, low = 1 << C['top'], 1 << C['low']
    assert C['top'] - C['low'] >= 2                                      # wider than one downward step plus one bit of noise
    for fan in (1, 32, 1700, 50000):
        for true_log2 in range(-6, 30):                                   # the largest final sum of the level, in units of the largest contribution
            adj, history = 0, []
            for call in range(64):
                h = headroom_bits(fan, adj)
                fm = int(2.0 ** (true_log2 + 31 - h))                      # largest |field| in units of this call
                fm = min(fm, (1 << 31) - 1)
                history.append((h, fm))
                new = headroom_feedback(adj, fm)
                if new == adj:
                    break
                adj = new
            h, fm = history[-1]
            settled = low <= fm < top
            pinned = (h == C['h_min'] and fm < low) or (h == C['h_max'] and fm >= top) or (adj == C['adj_floor'] and fm < low)
            assert settled or pinned, (fan, true_log2, history[-3:])
            if settled:
                # stays: the same gradient again changes nothing, and the call that corrected upward did not overshoot the lower edge
                assert headroom_feedback(adj, fm) == adj
            ups = [i for i in range(1, len(history)) if history[i][0] > history[i - 1][0]]
            downs = [i for i in range(1, len(history)) if history[i][0] < history[i - 1][0]]
            assert not (ups and downs), (fan, true_log2, history)         # no oscillation: a stationary gradient moves the unit ONE way
            # upward: the excess bits at once -- ONE correction from a value that was measured (a saturated int32 says nothing about the excess)
            assert sum(1 for i in ups if history[i - 1][1] < (1 << 31) - 1) <= 1, history
            for i in ups:
                if history[i - 1][1] < (1 << 31) - 1 and history[i][0] < C['h_max']:       # (not clamped at the widest headroom)
                    assert (1 << (C['top'] - 2)) <= history[i][1] < (1 << (C['top'] - 1)), history     # ... and it lands two bits below the top


def test_a_sixteenfold_jump_between_two_calls_is_not_flagged():
    margin = C['flag'] - C['top']
    assert margin == 4                                                   # 16x: DESIGN.md 5.1 (the soak flagged 4x jumps with a 2-bit margin)
    top = 1 << C['top']
    for fm in (1 << C['low'], top - 1):
        for jump in (2, 4, 8, 15.99):
            assert fm * jump < (1 << C['flag'])
    assert (top - 1) * 16.01 >= (1 << C['flag'])                          # ... and that is exactly where it ends


def test_the_resolution_that_is_left():
    # closed loop: the largest contribution of a call spans at least 2^(31 - h_max) units
    assert 31 - C['h_max'] >= 3
    # static guess (no state): never coarser than 2^-7 of the largest contribution, 64x the average fan-in before a flag
    assert 31 - C['h_max_static'] >= 7
    for fan in (1, 2, 32, 1700):
        h = headroom_bits(fan)
        assert (1 << h) >= min(64 * fan, 1 << C['h_max_static']) or h == C['h_min_static']
    # a fresh state starts on the safe side of the static guess by kHeadroomStartBias bits
    assert headroom_bits(32, 0) - ((31).bit_length() + 6) == C['bias'] > 0
