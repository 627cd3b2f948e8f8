"""Dev tool: where the sorted owners' time goes.  PERF_BWD_EXP bits: 1 no apply, 2 no gathers, 4 / 8 / 12: 1 / 2 / 8 records
per thread and step (default 4).    python tools/exp/bwd_sort_exp.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bwd_sort_lib import *

for kind in ('rays', 'random'):
    n = 1 << 20
    x, dfeat, amax = batch(kind, n)
    ws = None
    for sort, exp in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 3), (1, 4), (1, 8), (1, 12), (1, 5), (1, 13)):
        setenv(sort, 1); os.environ['PERF_BWD_EXP'] = str(exp)
        out, ws = call(x, dfeat, amax, ws=ws)
        ms = timed(lambda: call(x, dfeat, amax, ws=ws, out=out))
        call(x, dfeat, amax, ws=ws, out=out); torch.cuda.synchronize()
        rows = block_times(ws, n, bool(sort))
        hashed = [r[3] for r in rows if r[0] >= 4]
        print(kind, f'sort={sort} exp={exp:2d}: {ms:.4f} ms;  hashed owners mean {sum(hashed) / len(hashed):6.1f} us  (levels 4 / 8 / 15: {rows[4][3]:.0f} / {rows[8][3]:.0f} / {rows[15][3]:.0f});'
              f'  dense levels 0-3: ' + ' / '.join(f'{r[3]:.0f}' for r in rows[:4]), flush=True)
