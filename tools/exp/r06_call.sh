R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06cj
rm -rf $O; mkdir -p $O
cd $R
( PERF_COLOR_BESIDE_GRADIENT=1 timeout 1800 python -m pytest tests/test_gpu_scene.py tests/test_gpu_counts.py -m gpu -x -q -k "step or scene or train or graph or geo or episode" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 0 1 0 1; do
( PERF_COLOR_BESIDE_GRADIENT=$i timeout 900 python bench.py --no-config5 --no-config4 ) > $O/bench_$i.log 2> $O/bench_$i.err
python - <<PY
import json
t=open('$O/bench_$i.log').read()
d=json.loads([l for l in t.splitlines() if l.startswith('{')][-1])
print($i, d['ms_per_step'], d['faithful'], d['summary'].get('episode_psnr_db'), d['summary'].get('episode_train_seconds'))
PY
done
