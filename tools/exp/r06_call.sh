# Round 6 GPU call (rewritten per call; the log of calls is profiles/r06_gpurun_calls.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06b
rm -rf $O; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_runner_state.py tests/test_gpu_counts.py -m gpu -q -s --durations=8 ) > $O/pytest_new.log 2>&1
echo "pytest rc=$?" >> $O/pytest_new.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exp/c5_encode_probe.hip -o /tmp/c5_probe > $O/probe_build.log 2>&1
timeout 900 /tmp/c5_probe 28 30 > $O/c5_probe.jsonl 2> $O/c5_probe.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5_FETCH -o p -- /tmp/c5_probe 28 > $O/c5_probe_fetch.log 2>&1
cd $R
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; tail -12 $O/pytest_new.log; cat $O/c5_probe.jsonl
