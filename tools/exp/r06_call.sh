R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06w
rm -rf $O; mkdir -p $O
cd $R
for SC in room doorway pillars; do
  timeout 900 python tools/soak_episodes.py --episodes 25 --scene $SC > $O/soak_$SC.log 2>&1
  tail -1 $O/soak_$SC.log > $O/soak_$SC.json
done
timeout 900 python tools/shim_level_episode.py > $O/shim_level.log 2>&1
cp gpurun_out/shim_level_episode.json $O/ 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29711 tools/soak_episodes.py --one-device --episodes 5 --scene doorway --geo 300 --app 300 --height 256 --width 512 --batch 1024 --out $O/dp_soak.json > $O/dp_soak.log 2>&1
du -sh $O; for SC in room doorway pillars; do python -c "
import json; d=json.load(open('$O/soak_$SC.json')); print('$SC', d['psnr_min_max'], d['seconds_min_max'], d['skipped_for_overflow_total'], d['skipped_for_truncation_total'], [e['fp32_repairs_app_net'] for e in d['episodes']][-1])"; done; tail -3 $O/shim_level.log | cut -c1-300
