R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for F in 4096 32768 131072; do
  for i in 1 2; do
    PERF_FUSED_MAX_SAMPLES=$F timeout 300 python tools/train_episode.py 2>/dev/null > /tmp/ep.json
    F=$F python -c "
import json,os; d=json.load(open('/tmp/ep.json')); k=d['kernels_per_step (launches, us per launch)']; print(os.environ['F'], round(d['episode_s'],3), round(d['ms_per_geo_step'],4), round(d['ms_per_app_step'],4), round(d['psnr_dB'],2), k['geo_launches_per_step'], k['geo_sum_us_per_step'], k['app_launches_per_step'])"
  done
done
