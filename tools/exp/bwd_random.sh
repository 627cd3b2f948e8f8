#!/bin/bash
# grid backward on ray-ordered vs uniformly random points (the latter reach the last cell of the dense levels, whose corner indices wrap)
python - <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools', 'exp'))
from bwd_sort_lib import *
for kind in ('rays', 'random'):
    x, dfeat, amax = batch(kind, 1 << 20)
    out, ws = call(x, dfeat, amax)
    ms = timed(lambda: call(x, dfeat, amax, ws=ws, out=out))
    call(x, dfeat, amax, ws=ws, out=out); torch.cuda.synchronize()
    rows = block_times(ws, 1 << 20, False)
    print(kind, f'{ms:.4f} ms;', 'per-workgroup us by level:', ' '.join(f'{r[3]:.0f}' for r in rows))
PY
