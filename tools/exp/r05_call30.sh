R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05ab
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-200
timeout 600 python bench.py > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench_line.json
python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['ms_per_step'], d['value']); print(d['psnr']['train_seconds'], d['psnr']['curve']); print(d['faithful']['geo_ms_per_step'], d['faithful']['app_ms_per_step'], d['config4']['frame_as_one_batch']['frames_per_s'], d['train_app']['ms_per_step'])"
for i in 1 2; do timeout 300 python tools/soak_episodes.py --episodes 3 --scene room 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print([round(e['seconds'],3) for e in d['episodes']], [round(e['psnr_dB'],2) for e in d['episodes']], d['params_sha256_16'])"; done
