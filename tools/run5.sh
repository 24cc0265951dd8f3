cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/gputest_5.log 2>&1
grep "grad-check" gpurun_out/gputest_5.log | sort -t% -k2 | awk '{print}' > gpurun_out/grad_stats.txt
tail -4 gpurun_out/gputest_5.log; wc -l gpurun_out/grad_stats.txt
