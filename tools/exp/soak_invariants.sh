#!/bin/bash
# Options that must not change a single bit of the trained parameters: two full episodes (2 x 4,500 steps) per variant,
# sha256 of both parameter vectors after every episode (tools/soak_episodes.py).
mkdir -p gpurun_out/soak_inv
i=0
for f in "" "--eager" "--no-reuse" "--head 0" "--head 4" "--rccl-single-rank"; do
  python tools/soak_episodes.py --episodes 2 $f 2>/dev/null | grep '^{"params' > gpurun_out/soak_inv/v$i.json
  i=$((i+1))
done
python - <<'PY'
import json
names = ['default (hipGraph replays, features reused, two-phase sampler K = 2)', 'eager launches', 'strict two-encode order', 'one-phase sampler', 'two-phase sampler, K = 4',
         'data-parallel path on a single-rank RCCL world']
out = {'command': 'bash tools/exp/soak_invariants.sh', 'what': 'sha256 (first 16 hex digits) of the geometry + colour parameter vectors after each of two consecutive full episodes (3000 + 1500 iterations, 8192-ray batches, bf16)', 'variants': {}}
for i, n in enumerate(names):
    d = json.load(open(f'gpurun_out/soak_inv/v{i}.json'))
    out['variants'][n] = {'digests': [e['params_sha256_16'] for e in d['episodes']], 'seconds': [e['seconds'] for e in d['episodes']], 'psnr_dB': [e['psnr_dB'] for e in d['episodes']],
                          'skipped_steps': d['skipped_for_overflow_total'] + d['skipped_for_truncation_total']}
ref = out['variants'][names[0]]['digests']
out['all_equal'] = all(v['digests'] == ref for v in out['variants'].values())
json.dump(out, open('gpurun_out/soak_inv/summary.json', 'w'), indent=1)
print(out['all_equal'], {n[:30]: v['digests'] for n, v in out['variants'].items()})
PY
