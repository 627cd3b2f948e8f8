R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06di
rm -rf $O; mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -c 300 $O/bench.log; tail -3 $O/bench.err
