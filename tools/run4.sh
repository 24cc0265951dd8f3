cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity_r2.py -x -q -s -k "sharded_step or training_loss" ) > gpurun_out/shard_test.log 2>&1
grep -v "Warning\|warn\|^\s*$\|return func\|run_backward" gpurun_out/shard_test.log | tail -60
