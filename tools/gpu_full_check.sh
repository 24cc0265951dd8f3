cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputest_full.log 2>&1
grep -E "passed|failed|error" gpurun_out/gputest_full.log | tail -3
F="--steps 20 --warmup 5 --trace-steps 0 --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare"
P='import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], "ms_per_step", round(d["ms_per_step"],4), "value", round(d["value"]/1e6,1), d["config"].get("workload"), (d.get("north_star_sharded") or {}).get("ms_per_step"))'
python bench.py $F > gpurun_out/bf_head.json 2> gpurun_out/bf_head.err; python -c "$P" gpurun_out/bf_head.json
python bench.py $F --with-loss > gpurun_out/bf_loss.json 2> gpurun_out/bf_loss.err; python -c "$P" gpurun_out/bf_loss.json
python bench.py $F --encoder attention > gpurun_out/bf_attn.json 2> gpurun_out/bf_attn.err; python -c "$P" gpurun_out/bf_attn.json
python bench.py $F --workload S-icews14 --with-loss > gpurun_out/bf_i14.json 2> gpurun_out/bf_i14.err; python -c "$P" gpurun_out/bf_i14.json
python bench.py $F --workload S-icews0515 --with-loss > gpurun_out/bf_i0515.json 2> gpurun_out/bf_i0515.err; python -c "$P" gpurun_out/bf_i0515.json
TEMP_BENCH_FORCE_DIST=1 python bench.py $F > gpurun_out/bf_dist.json 2> gpurun_out/bf_dist.err; python -c "$P" gpurun_out/bf_dist.json
python bench.py > gpurun_out/bf_default.json 2> gpurun_out/bf_default.err; python -c "
import json
d=json.loads(open('gpurun_out/bf_default.json').read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['value'], d['config']['train_loop'])"
# N > 1 pre-flight on ONE GPU (verdict r5 item 8): eight processes, HIP kernels + gloo collectives, both shard modes; a functional
# check of the launch / exchange / all-reduce path (3-4 minutes), not a measurement
TEMP_BENCH_DIST_BACKEND=gloo timeout 1200 python bench.py --gpus 8 --steps 3 --warmup 1 --train-loop-steps 0 --no-cpu-baseline --no-extras --no-fp32-mfma-compare > gpurun_out/bf_gloo8.json 2> gpurun_out/bf_gloo8.err; python -c "
import json
d=json.loads(open('gpurun_out/bf_gloo8.json').read().strip().splitlines()[-1]); print('gloo x8', d['n_gpus'], d['ms_per_step'], (d.get('north_star_sharded') or {}).get('ms_per_step'))"
