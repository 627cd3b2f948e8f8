R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05g
rm -rf $O; mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 --durations=8 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python bench.py ) > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log > $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ep_kt -o ep -- python $R/tools/train_episode.py --geo 1000 --app 500 > $O/ep_kt.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -15 $O/pytest.log | cut -c1-300; python -c "
import json; d=json.load(open('$O/bench_line.json'))
print('ms_per_step', d['ms_per_step'], 'faithful', d['faithful'], 'psnr train s', d['psnr']['train_seconds'], 'config4', d['config4']['frame_as_one_batch']['frames_per_s'], 'train_app', d['train_app']['ms_per_step'])
print({k: (v['seconds_per_panorama'], v['roofline']['frac'], v['roofline']['moved_frac']) for k, v in d['config5'].items()})
"
