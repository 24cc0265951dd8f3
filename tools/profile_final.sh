cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_round.sh r03
python tools/step_sequence.py gpurun_out/prof_r03/bench_kernel_trace.csv 40 > gpurun_out/r03_step_sequence.txt 2>&1; tail -3 gpurun_out/r03_step_sequence.txt
for R in 230 20; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_shbm${R}_$c -o pmc -- python bench.py --workload S-hbm --shbm-relations $R --steps 2 --warmup 1 --trace-steps 0 > gpurun_out/pmc_shbm${R}_$c.log 2>&1
  done
  f=$(find gpurun_out/pmc_shbm${R}_FETCH_SIZE -name '*counter_collection.csv' | head -1)
  w=$(find gpurun_out/pmc_shbm${R}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  python tools/pmc_summary.py $f $w gpurun_out/pmc_traffic_shbm_$R.json > gpurun_out/pmc_shbm_$R.txt 2>&1
  head -8 gpurun_out/pmc_shbm_$R.txt
  rm -rf gpurun_out/pmc_shbm${R}_FETCH_SIZE gpurun_out/pmc_shbm${R}_WRITE_SIZE
done
