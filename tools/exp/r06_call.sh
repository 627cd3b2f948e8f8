# Round 6 GPU call (rewritten per call; the log of calls is profiles/r06_gpurun_calls.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p
rm -rf $O; mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_scene.py tests/test_gpu_counts.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
for T in three_a one_a three_b one_b; do
  F=""; case $T in three*) F="--three-calls";; esac
  timeout 600 python tools/shim_step_profile.py --steps 200 $F --out $O/shim_$T > $O/shim_$T.log 2>&1
done
rm -f $O/*trace.json
du -sh $O; tail -4 $O/pytest.log; python -c "
import json
for t in ('three_a','one_a','three_b','one_b'):
    d=json.load(open('$O/shim_%s_host.json' % t)); print(t, d['geo_ms_per_step'], d['app_ms_per_step'], d['geo']['host_self_cpu_us_per_step'].get('_FieldFnBackward'), d['geo']['host_self_cpu_us_per_step'].get('hipLaunchKernel'))
"
