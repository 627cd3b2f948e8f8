# VERDICT-4 #1d: the next step's draw + marching on a second stream under data parallelism (single-rank RCCL world)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05e
rm -rf $O; mkdir -p $O
cd $R
OFF="--no-cpu-baseline --no-psnr --no-reuse-line --no-render-block --no-config4 --no-config5 --no-train-app --sustain-seconds 0.5"
for M in 0 1 2; do
  PERF_DP_SINGLE_RANK=1 PERF_PIPELINE_MARCHING=$M PERF_PIPELINE_MARCHING_DP=1 timeout 300 python bench.py $OFF > $O/dp_pm$M.log 2> $O/dp_pm$M.err
  tail -1 $O/dp_pm$M.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipeline', $M, 'ms_per_step', d['ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'comm', {k: d['comm'].get(k) for k in ('single_rank_step_ms','exposed_comm_ms')}, d['config']['launch'])"
done
