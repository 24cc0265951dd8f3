cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/tile_phases.py > gpurun_out/tile_phases.log 2>&1; cat gpurun_out/tile_phases.log | grep -v Warn | grep "VAR\|main"
python bench.py --steps 20 --warmup 5 --kernel-table --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare > gpurun_out/bench_2.json 2> gpurun_out/bench_2.err
grep "k_rgcn\|k_fixup" gpurun_out/bench_2.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
