cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputest_15.log 2>&1
grep -E "passed|failed|error" gpurun_out/gputest_15.log | tail -3
python bench.py --steps 20 --warmup 5 --trace-steps 0 --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare > gpurun_out/bench_15.json 2> gpurun_out/bench_15.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_15.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['value'])"
python bench.py --workload S-hbm --steps 5 --warmup 2 > gpurun_out/b15_shbm.json 2> gpurun_out/b15_shbm.err
python bench.py --workload S-hbm --shbm-relations 20 --steps 5 --warmup 2 > gpurun_out/b15_shbm20.json 2> gpurun_out/b15_shbm20.err
python - <<'PY'
import json
for f in ('gpurun_out/b15_shbm.json','gpurun_out/b15_shbm20.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], {k:(round(v['avg_ms'],3)) for k,v in d['kernels'].items()})
PY
