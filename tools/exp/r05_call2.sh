# Round 5, second GPU call: the whole GPU suite (no -x), the shim-level step again (descriptor cache), smoke.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05b
rm -rf $O; mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 --durations=25 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python tools/shim_step_profile.py --steps 200 --out $O/shim > $O/shim.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
rm -f $O/*_trace.json
tail -30 $O/pytest.log; tail -12 $O/shim.log; tail -3 $O/smoke.log
