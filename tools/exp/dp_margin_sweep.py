"""Lagged fixed-point units of the sharded exchange: steps the job-wide gate skips for an overflow flag, and PSNR@iter against the
oracle curve, as a function of the margin (bits) the lagged units add on top of the previous step's statistics.
   python tools/exp/dp_margin_sweep.py 1 2 3 4"""
import json, os, subprocess, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
golden = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'psnr_curve.json')))
rows = {str(r['seed']): r['oracle'] for r in golden['seeds']}
seeds = [str(r['seed']) for r in golden['seeds']][:3]
out = {}
for m in sys.argv[1:]:
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    path = f'/tmp/dp_margin_{m}.json'
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT, PERF_DP_LAG_MARGIN_BITS=m)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'tests', 'psnr_dp_worker.py'), path] + seeds, env=env, capture_output=True, text=True)
    if os.environ.get('PERF_DP_DIAG') == '1':
        print('\n'.join(l for l in r.stdout.splitlines() if 'DIAG' in l or l.startswith('   '))[:6000])
    if r.returncode != 0:
        out[m] = r.stderr[-500:]; continue
    res = json.load(open(path))['curves']
    out[m] = {'skipped': {s: res[s]['skipped_for_overflow'] for s in seeds}, 'which': {s: res[s]['steps_skipped_for_overflow'] for s in seeds},
              'dpsnr150': [round(res[s]['psnr@app150'] - rows[s]['psnr@app150'], 3) for s in seeds],
              'dpsnr300': [round(res[s]['psnr@app300'] - rows[s]['psnr@app300'], 3) for s in seeds]}
    print(m, json.dumps(out[m]), flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'dp_margin_sweep.json'), 'w'), indent=1)
