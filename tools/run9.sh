cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputest_9.log 2>&1
tail -6 gpurun_out/gputest_9.log
python bench.py --steps 20 --warmup 5 --kernel-table --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare > gpurun_out/bench_9.json 2> gpurun_out/bench_9.err
grep "k_gemm_panel" gpurun_out/bench_9.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_9.json').read().strip().splitlines()[-1]); print('resident on ', d['ms_per_step'], d['value'])"
TEMP_GEMM_RESIDENT=0 python bench.py --steps 20 --warmup 5 --trace-steps 0 --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare > gpurun_out/bench_9b.json 2> gpurun_out/bench_9b.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_9b.json').read().strip().splitlines()[-1]); print('resident off', d['ms_per_step'], d['value'])"
