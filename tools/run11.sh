cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q -n 4 ) > gpurun_out/gputest_11.log 2>&1
grep -E "passed|failed" gpurun_out/gputest_11.log | tail -3
python bench.py --steps 20 --warmup 5 --kernel-table --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare > gpurun_out/bench_11.json 2> gpurun_out/bench_11.err
grep -E "k_gemm_panel|k_gemm_tn|k_gru_chain" gpurun_out/bench_11.err | head
python -c "
import json
d=json.loads(open('gpurun_out/bench_11.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['value'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_11 -o bench -- python bench.py --steps 10 --warmup 3 --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare --trace-steps 0 > gpurun_out/prof_11_line.json 2> gpurun_out/prof_11.err
python tools/step_sequence.py gpurun_out/prof_11/bench_kernel_trace.csv 40 2>&1 | grep -E "bxr|bxp|period" | cut -c1-120
