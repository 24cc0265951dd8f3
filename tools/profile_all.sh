#!/bin/bash
# GPU box: every committed measurement of a round in ONE gpurun call; results under gpurun_out/final_<tag>/ (copied to profiles/
# by hand afterwards).  Order matters: the counter passes first -- bench.py reads profiles/<tag>_pmc_traffic*.json for the
# (static-marked) traffic fields of the lines taken after them.
tag=${1:-r05}
export TMPDIR=/tmp
out=gpurun_out/final_$tag
mkdir -p $out
cmd="rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python bench.py --steps 20 --warmup 5 --train-loop-steps 0 --no-fp32-mfma-compare --no-extras"
bash tools/profile_round.sh $tag > $out/profile_round.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${tag}_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_${tag}_WRITE_SIZE/pmc_counter_collection.csv profiles/${tag}_pmc_traffic.json > $out/pmc_summary.txt 2>&1
python tools/profile_md.py gpurun_out/prof_$tag/bench_kernel_stats.csv gpurun_out/prof_${tag}_line.json $tag "$cmd" auto > $out/profile_md.txt 2>&1
python tools/step_sequence.py gpurun_out/prof_$tag/bench_kernel_trace.csv > profiles/${tag}_step_sequence.txt 2>$out/step_sequence.err
bash tools/profile_hbm_window.sh $tag > $out/profile_hbmw.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${tag}_hbmw_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_${tag}_hbmw_WRITE_SIZE/pmc_counter_collection.csv profiles/${tag}_pmc_traffic_hbm_window.json > $out/pmc_summary_hbmw.txt 2>&1
cp gpurun_out/prof_${tag}_hbmw/bench_kernel_stats.csv profiles/${tag}_hbm_window_kernel_stats.csv
python bench.py --workload S-hbm-window > $out/hbm_window_line.json 2> $out/hbm_window.err
bash tools/profile_shbm.sh > $out/profile_shbm.log 2>&1
for r in 230 20; do
  cp gpurun_out/prof_shbm$r/bench_kernel_stats.csv profiles/${tag}_shbm_rel${r}_kernel_stats.csv
  cp gpurun_out/prof_shbm${r}_line.json profiles/${tag}_shbm_rel${r}_line.json
done
python bench.py --steps 20 --warmup 5 > $out/driver_cmd_line.json 2> $out/driver_cmd.err
python bench.py > $out/default_line.json 2> $out/default.err
timeout 300 python tools/stack_probe.py > $out/stack_probe.txt 2>&1
timeout 300 python tools/stack_probe.py --loop > $out/stack_probe_loop.txt 2>&1
timeout 300 python tools/generic_probe.py 2>&1 | head -4 > $out/generic_probe.txt
# what was (re)written under profiles/ on the box travels back through gpurun_out/
mkdir -p $out/profiles
cp profiles/${tag}_* $out/profiles/
rm -rf gpurun_out/pmc_${tag}_* gpurun_out/prof_${tag}_hbmw gpurun_out/prof_shbm230 gpurun_out/prof_shbm20
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
du -sh gpurun_out
tail -c 400 $out/default_line.json
