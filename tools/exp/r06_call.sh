R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06cz
rm -rf $O $R/gpurun_out/r06; mkdir -p $O
cd $R
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -c 1200 $O/bench.log; tail -3 $O/bench.err
bash tools/exp/r06_profile.sh > $O/profile.log 2>&1; tail -5 $O/profile.log
