#!/bin/bash
# Options that must not change a single bit of the trained parameters: two full episodes (2 x 4,500 steps) per variant,
# sha256 of both parameter vectors after every episode (tools/soak_episodes.py).  Round 4: the data-parallel variant runs with
# EXACT units (the bit-identical mode); the default lagged units and the other marching lattice are reported beside the set.
mkdir -p gpurun_out/soak_inv
i=0
for f in "" "--eager" "--no-reuse" "--head 0" "--head 4" "EXACT --rccl-single-rank" "PIPE" "NOIDX --head 0"; do
  env_=""
  case "$f" in EXACT*) env_="PERF_DP_UNITS=exact"; f="${f#EXACT }";; PIPE) env_="PERF_PIPELINE_MARCHING=1"; f="";; NOIDX*) env_="PERF_INDEX_FEATURES=0"; f="${f#NOIDX }";; esac
  env $env_ python tools/soak_episodes.py --episodes 2 $f 2>/dev/null | grep '^{"params' > gpurun_out/soak_inv/v$i.json
  i=$((i+1))
done
PERF_DP_UNITS=lagged python tools/soak_episodes.py --episodes 2 --rccl-single-rank 2>/dev/null | grep '^{"params' > gpurun_out/soak_inv/lagged.json
python - <<'PY'
import json
names = ['default (hipGraph replays, features reused, two-phase sampler K = 2, repeated-addition lattice, repair launch)', 'eager launches', 'strict two-encode order', 'one-phase sampler',
         'two-phase sampler, K = 4', 'data-parallel path on a single-rank RCCL world, exact units', 'pipelined marching (second stream beside the backward)',
         'one-phase sampler, features of the kept samples COPIED at compaction (PERF_INDEX_FEATURES=0) instead of read through rows']
out = {'command': 'bash tools/exp/soak_invariants.sh', 'what': 'sha256 (first 16 hex digits) of the geometry + colour parameter vectors after each of two consecutive full episodes (3000 + 1500 iterations, 8192-ray batches, bf16)', 'variants': {}}
def row(d):
    return {'digests': [e['params_sha256_16'] for e in d['episodes']], 'seconds': [e['seconds'] for e in d['episodes']], 'psnr_dB': [e['psnr_dB'] for e in d['episodes']],
            'flagged_steps': d['skipped_for_overflow_total'], 'truncated_steps': d['skipped_for_truncation_total']}
for i, n in enumerate(names):
    out['variants'][n] = row(json.load(open(f'gpurun_out/soak_inv/v{i}.json')))
ref = out['variants'][names[0]]['digests']
out['all_equal'] = all(v['digests'] == ref for v in out['variants'].values())
out['beside the set (NOT bit-identical by design)'] = {'data-parallel path, lagged units (the default exchange)': row(json.load(open('gpurun_out/soak_inv/lagged.json')))}
json.dump(out, open('gpurun_out/soak_inv/summary.json', 'w'), indent=1)
print(out['all_equal'], {n[:30]: v['digests'] for n, v in out['variants'].items()})
PY
