# Final artefacts of a round on the GPU box (one gpurun call): GPU tests, smoke, the bench line, config 4, the training episode,
# kernel traces and counter passes.  Everything lands in gpurun_out/final/ (+ gpurun_out/r03/, gpurun_out/ep_prof/); the folds
# into profiles/ are done afterwards by tools/exp/r03_fold.py and by hand-copying the JSON lines.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/final; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=20 > $O/tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench_line.json 2> $O/bench.err
python tools/render_dense.py > $O/render_dense_16.json 2> $O/rd.err
python tools/render_dense.py --batch 524288 > $O/render_dense_frame.json 2>> $O/rd.err
python tools/train_episode.py > $O/train_episode.json 2> $O/ep.err
python tools/exp/fwd_v2.py --out $O/fwd_v2.json > $O/fwd_v2.log 2>&1
bash tools/exp/dp_single.sh > $O/dp_single.log 2>&1
bash tools/exp/ep_prof.sh > $O/ep_prof.log 2>&1
bash tools/exp/r03_profile.sh > $O/r03_profile.log 2>&1
grep -E "passed|failed" $O/tests.log | tail -2; tail -1 $O/smoke.log; python - <<'P'
import json
d=json.load(open('gpurun_out/final/bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['bound'], d['sustained']['value'])
P
