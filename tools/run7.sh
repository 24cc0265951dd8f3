cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py -x -q -k "rgcn or tiled or determinism or layer or sharded" ) > gpurun_out/t7.log 2>&1
tail -4 gpurun_out/t7.log
python bench.py --steps 20 --warmup 5 --kernel-table --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare > gpurun_out/bench_7.json 2> gpurun_out/bench_7.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_7.json').read().strip().splitlines()[-1]); print('overlap on ', d['ms_per_step'], d['value'])"
TEMP_OVERLAP=0 python bench.py --steps 20 --warmup 5 --trace-steps 0 --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare > gpurun_out/bench_7b.json 2> gpurun_out/bench_7b.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_7b.json').read().strip().splitlines()[-1]); print('overlap off', d['ms_per_step'], d['value'])"
for R in 230 20; do python bench.py --workload S-hbm --shbm-relations $R --steps 5 --warmup 2 > gpurun_out/shbm_$R.json 2> gpurun_out/shbm_$R.err; done
