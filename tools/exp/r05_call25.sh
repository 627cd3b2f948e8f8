cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05y
mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "mlp" > $O/t1.log 2>&1; tail -5 $O/t1.log; cd /tmp
OFF="--no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0 --no-render-block --no-config4 --no-config5 --no-train-app"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 $OFF > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_app -o kt -- python $R/bench.py --steps 20 --warmup 5 --mode train_app $OFF > $O/kt_app.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
grep -h "mlp_bwd" $O/kt/kt_kernel_stats.csv $O/kt_app/kt_kernel_stats.csv | cut -c1-60,200-400
tail -1 $O/kt.log | cut -c1-200
