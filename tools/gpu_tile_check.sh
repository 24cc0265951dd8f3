cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -k "tile or determinism or rgcn or layer or full_size or golden_gpu" ) > gpurun_out/gputest_tile.log 2>&1
grep -E "passed|failed" gpurun_out/gputest_tile.log | tail -3
python tools/tile_phases.py 2>&1 | grep -v Warn | sed -n 1,12p
python bench.py --steps 20 --warmup 5 --kernel-table --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare > gpurun_out/bench_tile.json 2> gpurun_out/bench_tile.err
grep -E "k_rgcn|k_gemm_tn " gpurun_out/bench_tile.err | head
python -c "
import json
d=json.loads(open('gpurun_out/bench_tile.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['value'])"
