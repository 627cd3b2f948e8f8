# Round 5, first GPU call: the new tests + the whole GPU suite, the default bench line (config5 / train_app blocks), the
# shim-level step's host timeline + kernel trace, the config-5 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05a
rm -rf $O; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python bench.py ) > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log > $O/bench_line.json
timeout 600 python tools/shim_step_profile.py --steps 200 --out $O/shim > $O/shim.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/shim_kt -o kt -- python $R/tools/shim_step_profile.py --steps 200 --no-profiler --out $O/shim_kt_run > $O/shim_kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/c5_$C -o c -- python $R/tools/config5.py --pano-log2 28 30 --pano-batches 8 > $O/c5_$C.log 2>&1
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; tail -5 $O/pytest.log; tail -c 600 $O/bench_line.json; tail -8 $O/shim.log
