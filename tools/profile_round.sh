#!/bin/bash
# GPU box: the round's committed measurements (run from the repo root through gpurun).
#   1. rocprofv3 kernel-trace + stats of the headline bench (the CPU baseline and the training-loop probe are left in: the stats
#      list every kernel of the process; the per-step table in profiles/ divides by the encoder steps of the headline part only)
#   2. FETCH_SIZE and WRITE_SIZE in their own counter-only passes (eager launches, 4 encoder steps)
tag=${1:-r02}
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python bench.py --steps 20 --warmup 5 --train-loop-steps 0 --no-fp32-mfma-compare --no-extras > gpurun_out/prof_${tag}_line.json 2> gpurun_out/prof_${tag}.err
tail -c 600 gpurun_out/prof_${tag}_line.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -o pmc -- python bench.py --steps 3 --warmup 1 --no-graph --trace-steps 0 --no-cpu-baseline --train-loop-steps 0 --no-fp32-mfma-compare --no-extras > gpurun_out/pmc_${tag}_$c.log 2>&1
done
find gpurun_out/prof_$tag gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE -type f | head -20
du -sh gpurun_out/prof_$tag gpurun_out/pmc_${tag}_*
