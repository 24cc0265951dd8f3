cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -k "gemm or linear or layer or rgcn or full_size or golden_gpu" ) > gpurun_out/gputest_18.log 2>&1
grep -E "passed|failed" gpurun_out/gputest_18.log | tail -3
python bench.py --steps 20 --warmup 5 --kernel-table --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare > gpurun_out/bench_18.json 2> gpurun_out/bench_18.err
grep -E "k_gemm_panel|k_gemm_tn " gpurun_out/bench_18.err | head -8
python -c "
import json
d=json.loads(open('gpurun_out/bench_18.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['value'])"
