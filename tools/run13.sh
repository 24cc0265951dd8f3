cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
F="--steps 20 --warmup 5 --trace-steps 0 --train-loop-steps 0 --no-cpu-baseline --no-fp32-mfma-compare"
P='import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], "ms_per_step", round(d["ms_per_step"],4), "value", round(d["value"]/1e6,1), d["config"].get("workload"), json.dumps(d.get("north_star_sharded"))[:600])'
python bench.py $F > gpurun_out/b13_head.json 2> gpurun_out/b13_head.err; python -c "$P" gpurun_out/b13_head.json
python bench.py $F --with-loss > gpurun_out/b13_loss.json 2> gpurun_out/b13_loss.err; python -c "$P" gpurun_out/b13_loss.json
python bench.py $F --encoder attention > gpurun_out/b13_attn.json 2> gpurun_out/b13_attn.err; python -c "$P" gpurun_out/b13_attn.json
python bench.py $F --workload S-icews14 --with-loss > gpurun_out/b13_i14.json 2> gpurun_out/b13_i14.err; python -c "$P" gpurun_out/b13_i14.json
python bench.py $F --workload S-icews0515 --with-loss > gpurun_out/b13_i0515.json 2> gpurun_out/b13_i0515.err; python -c "$P" gpurun_out/b13_i0515.json
TEMP_BENCH_FORCE_DIST=1 python bench.py $F > gpurun_out/b13_dist.json 2> gpurun_out/b13_dist.err; python -c "$P" gpurun_out/b13_dist.json
python bench.py --workload S-hbm --steps 5 --warmup 2 > gpurun_out/b13_shbm.json 2> gpurun_out/b13_shbm.err; tail -c 1500 gpurun_out/b13_shbm.json
python bench.py --workload S-hbm --shbm-relations 20 --steps 5 --warmup 2 > gpurun_out/b13_shbm20.json 2> gpurun_out/b13_shbm20.err; tail -c 1500 gpurun_out/b13_shbm20.json
