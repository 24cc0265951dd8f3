#!/bin/bash
# GPU box: the HBM-regime window (bench.py --workload S-hbm-window = extra.hbm_window of the default line): rocprofv3 kernel stats
# and FETCH_SIZE / WRITE_SIZE in their own counter-only passes (2 timed + 1 warm-up + 1 traced = 4 steps per process).
tag=${1:-r05}
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_hbmw -o bench -- python bench.py --workload S-hbm-window --hbm-window-steps 2 > gpurun_out/prof_${tag}_hbmw_line.json 2> gpurun_out/prof_${tag}_hbmw.err
tail -c 400 gpurun_out/prof_${tag}_hbmw_line.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_hbmw_$c -o pmc -- python bench.py --workload S-hbm-window --hbm-window-steps 2 > gpurun_out/pmc_${tag}_hbmw_$c.log 2>&1
done
du -sh gpurun_out/prof_${tag}_hbmw gpurun_out/pmc_${tag}_hbmw_*
