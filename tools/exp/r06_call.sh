# Round 6 GPU call (rewritten per call; the log of calls is profiles/r06_gpurun_calls.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06q
rm -rf $O; mkdir -p $O
cd $R
for LAY in tcnn line_local; do
  ( timeout 600 python tools/config5.py --pano-log2 28 30 --layout $LAY ) > $O/c5_$LAY.log 2>&1
  echo "== $LAY"; grep -h "ray_samples_per_s\|\"frac\"\|ms_per_launch\|seconds_per_panorama" $O/c5_$LAY.log
done
