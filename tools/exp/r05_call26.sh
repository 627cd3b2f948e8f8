R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05z
mkdir -p $O
cd $R
for S in pillars doorway; do
  PERF_GRID_GRAD_ACCUM=fp32 timeout 600 python tools/soak_episodes.py --episodes 5 --scene $S > $O/soak_fp32_$S.log 2>&1
  tail -1 $O/soak_fp32_$S.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scene'], [round(e['psnr_dB'],2) for e in d['episodes']], [round(e['seconds'],2) for e in d['episodes']], [e['grid_gradient_mode'] for e in d['episodes']][:1])"
done
