"""Dev tool: the grid backward's owner variants (PERF_BWD_RUNS; PERF_BWD_SORT / PERF_BWD_BITMAP in the commits that had them) -- equality of the fixed-point gradients,
ms per call, per-workgroup times by level.    python tools/exp/bwd_sort.py [out.json]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bwd_sort_lib import *

res = {}
for kind in ('rays', 'random'):
    n = 1 << 20
    x, dfeat, amax = batch(kind, n)
    setenv(0, 0)
    ref, ws = call(x, dfeat, amax)
    ref = ref.clone()
    ref32, _ = call(x, dfeat, amax, fixed=False)
    ref32 = ref32.clone()
    res[kind] = {}
    for sort, runs in ((0, 0), (1, 0), (0, 1), (1, 1)):
        setenv(sort, runs)
        out, ws = call(x, dfeat, amax, ws=ws)
        eq = bool(torch.equal(out, ref))
        o32, _ = call(x, dfeat, amax, fixed=False, ws=ws)
        err32 = float((o32 - ref32).abs().max() / ref32.abs().max())
        ms = timed(lambda: call(x, dfeat, amax, ws=ws, out=out))
        call(x, dfeat, amax, ws=ws, out=out); torch.cuda.synchronize()
        rows = block_times(ws, n, bool(sort))
        res[kind][f'bitmap{sort}_runs{runs}'] = {'equal': eq, 'fp32_rel_err': err32, 'ms': ms, 'levels': rows}
        print(kind, f'bitmap={sort} runs={runs}: equal={eq} fp32 err {err32:.2e}  {ms:.4f} ms', flush=True)
        for r in rows:
            print('    level %2d tiles %2d x rep %2d: mean %7.1f us  max %7.1f us' % r)
    # live count below the capacity, small batches
    setenv(0, 0)
    nd = torch.tensor([700001], dtype=torch.int64, device=dev)
    r2, _ = call(x, dfeat, amax, n_dev=nd); r2 = r2.clone()
    setenv(1, 1)
    o2, _ = call(x, dfeat, amax, n_dev=nd)
    print(kind, 'n_dev=700001 equal:', bool(torch.equal(o2, r2)), flush=True)
    res[kind]['n_dev_equal'] = bool(torch.equal(o2, r2))
    for m in (5, 1000, 4097, 70001):
        xs, ds, am = batch(kind, max(128, (m + 127) // 128 * 128), seed=m)
        xs, ds = xs[:m].contiguous(), ds[:, :m].contiguous()
        setenv(0, 0)
        a, _ = call(xs, ds, am); a = a.clone()
        setenv(1, 1)
        b, _ = call(xs, ds, am)
        print(kind, f'n={m} equal:', bool(torch.equal(a, b)), flush=True)
        res[kind][f'small_{m}_equal'] = bool(torch.equal(a, b))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], 'w'), indent=1)
