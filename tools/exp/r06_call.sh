# Round 6 GPU call (rewritten per call; the log of calls is profiles/r06_gpurun_calls.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06n
rm -rf $O; mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python tools/shim_step_profile.py --steps 200 --out $O/shim > $O/shim.log 2>&1
( time timeout 900 python bench.py --no-config5 ) > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log > $O/bench_line.json
du -sh $O; tail -6 $O/pytest.log; tail -5 $O/shim.log; python -c "
import json
d=json.load(open('$O/bench_line.json'))
print(d['value'], d['ms_per_step']); print(d['faithful']); print(d['psnr']['train_seconds'])
"
