cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -k "sliced or tiled" ) > gpurun_out/gputest_14.log 2>&1
tail -15 gpurun_out/gputest_14.log
python bench.py --workload S-hbm --steps 5 --warmup 2 > gpurun_out/b14_shbm.json 2> gpurun_out/b14_shbm.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b14_shbm.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
for k,v in d['config']['kernels'].items() if 'kernels' in d.get('config',{}) else []:
    print(k, v)
print(json.dumps(d)[:3000])
PY
TEMP_RGCN_SLICE=0 python bench.py --workload S-hbm --steps 5 --warmup 2 > gpurun_out/b14_shbm_off.json 2> gpurun_out/b14_shbm_off.err
python -c "
import json
d=json.loads(open('gpurun_out/b14_shbm_off.json').read().strip().splitlines()[-1]); print('slice off: ms_per_step', d['ms_per_step'])"
