R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06ba
rm -rf $O; mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests/test_gpu_config4.py tests/test_gpu_scene.py tests/test_gpu_runner_state.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
timeout 600 python tools/render_dense.py --poses 300 --batch 524288 > $O/rd_$i.log 2>&1
python - <<PY
import json
t=open('$O/rd_$i.log').read()
d=json.loads(t[t.rindex('\n{'):] if '\n{' in t else t[t.index('{'):])
print({k: d[k] for k in d if 'frames_per_s' in k or 'checksum' in k})
PY
done
