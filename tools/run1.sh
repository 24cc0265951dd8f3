cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputest_1.log 2>&1
tail -5 gpurun_out/gputest_1.log
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_seq -o seq -- python bench.py --steps 2 --warmup 1 --no-graph --trace-steps 0 --no-cpu-baseline --train-loop-steps 0 --no-fp32-mfma-compare > gpurun_out/trace_seq.log 2>&1
f=$(find gpurun_out/trace_seq -name '*kernel_trace.csv' | head -1)
python tools/step_sequence.py $f 30 > gpurun_out/step_sequence.txt 2>&1
tail -3 gpurun_out/step_sequence.txt
rm -rf gpurun_out/trace_seq
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_1.json 2> gpurun_out/bench_1.err
tail -c 1500 gpurun_out/bench_1.json
