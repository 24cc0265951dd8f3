cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputest_10.log 2>&1
tail -6 gpurun_out/gputest_10.log
python bench.py > gpurun_out/bench_10.json 2> gpurun_out/bench_10.err
tail -c 3000 gpurun_out/bench_10.json
bash tools/profile_round.sh r03
