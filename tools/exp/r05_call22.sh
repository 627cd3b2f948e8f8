R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05v
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_counts.py -m gpu -q -x > $O/t1.log 2>&1; tail -3 $O/t1.log
timeout 1500 python -m pytest tests/test_gpu_scene.py tests/test_gpu_psnr.py tests/test_gpu_config4.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=5 -k "not pillars" > $O/t2.log 2>&1; tail -8 $O/t2.log
timeout 600 python bench.py --no-cpu-baseline --no-config5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value']); b=d.get('blocks',d)
for k in ('faithful','config4'): print(k, json.dumps(b.get(k))[:700])
print(json.dumps(b.get('psnr'))[:400])"
