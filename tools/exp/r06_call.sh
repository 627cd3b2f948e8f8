R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06co
rm -rf $O; mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -c 1500 $O/bench.log; tail -4 $O/bench.err
cd /tmp && export TMPDIR=/tmp
OFF="--no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0 --no-render-block --no-config4 --no-config5 --no-train-app"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 $OFF > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_app -o kt -- python $R/bench.py --steps 20 --warmup 5 --mode train_app $OFF > $O/kt_app.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ep -o kt -- python $R/tools/train_episode.py > $O/kt_ep.log 2>&1
cd $R
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
find $O -name "*.csv" | head
