R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05ac
mkdir -p $O
cd $R
for S in room doorway pillars; do
  timeout 900 python tools/soak_episodes.py --episodes 25 --scene $S > $O/soak25_$S.log 2>&1
  tail -1 $O/soak25_$S.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scene'], d['psnr_min_max'], d['seconds_min_max'], d['skipped_for_overflow_total'], d['skipped_for_truncation_total'], d['mem_reserved_MB_first_last'])"
done
