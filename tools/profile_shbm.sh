#!/bin/bash
# GPU box: rocprofv3 kernel stats of the S-hbm layer bench (230 and 20 relations) -> gpurun_out/prof_shbm*; summaries are copied
# to profiles/ by hand (kernel_stats.csv of each run).
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 230 20; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_shbm$r -o bench -- python bench.py --workload S-hbm --shbm-relations $r --steps 10 --warmup 2 > gpurun_out/prof_shbm${r}_line.json 2> gpurun_out/prof_shbm$r.err
  tail -c 300 gpurun_out/prof_shbm${r}_line.json
done
find gpurun_out/prof_shbm230 gpurun_out/prof_shbm20 -name "*stats*" | head
