# Round 5, final evidence run (one gpurun call): GPU suite, smoke, the default bench line, the data-parallel line on a single-rank RCCL
# world, kernel traces + counter passes (tools/exp/r05_profile.sh), soaks per scene family, the shim-level episode.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 --durations=10 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
( time timeout 600 python bench.py ) > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log > $O/bench_line.json
PERF_DP_SINGLE_RANK=1 timeout 600 python bench.py --no-cpu-baseline --no-psnr --no-render-block --no-config4 --no-config5 --no-train-app > $O/bench_dp1.log 2> $O/bench_dp1.err
tail -1 $O/bench_dp1.log > $O/bench_dp1_line.json
for S in room doorway pillars; do
  timeout 600 python tools/soak_episodes.py --episodes 5 --scene $S > $O/soak_$S.log 2>&1
done
timeout 600 python tools/soak_episodes.py --episodes 3 --scene room --no-shrink > $O/soak_room_noshrink.log 2>&1
timeout 600 python tools/shim_level_episode.py > $O/shim_level_episode.json 2> $O/shim_level_episode.err
bash tools/exp/r05_profile.sh > $O/profile.log 2>&1
tail -4 $O/pytest.log | cut -c1-200; tail -1 $O/smoke.log; for S in room doorway pillars; do tail -1 $O/soak_$S.log | cut -c1-400; done; tail -1 $O/soak_room_noshrink.log | cut -c1-200
