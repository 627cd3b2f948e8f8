R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06cp
rm -rf $O; mkdir -p $O
cd $R
( timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_counts.py tests/test_gpu_scene.py -m gpu -x -q -k "bwd or backward or grad or step or field or overflow or book" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for i in 1 2; do
( timeout 900 python bench.py --no-config5 --no-config4 ) > $O/bench_$i.log 2> $O/bench_$i.err
python - <<PY
import json
t=open('$O/bench_$i.log').read()
d=json.loads([l for l in t.splitlines() if l.startswith('{')][-1])
f=d['faithful']
print($i, d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'], f['geo_ms_per_step'], f['app_ms_per_step'], d['summary'].get('episode_psnr_db'), d['train_app']['ms_per_step'], d['kernels_late'].get('perf_hashgrid_bwd'))
PY
done
cd /tmp && export TMPDIR=/tmp
OFF="--no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0 --no-render-block --no-config4 --no-config5 --no-train-app"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 $OFF > $O/kt.log 2>&1
head -4 $O/kt/kt_kernel_stats.csv | cut -c1-60,200-400
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
