cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "A" "TEMP_OVERLAP=0" "TEMP_DEBUG=9" "TEMP_OVERLAP=0 TEMP_DEBUG=9"; do
  if [ "$v" = "A" ]; then e=""; else e="$v"; fi
  env $e python bench.py --workload S-hbm --steps 5 --warmup 2 > gpurun_out/b16.json 2> gpurun_out/b16.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/b16.json').read().strip().splitlines()[-1])
print(sys.argv[1], round(d['ms_per_step'],3), {k:(round(v['avg_ms'],3)) for k,v in d['kernels'].items() if 'rgcn' in k or 'fixup' in k})
PY
done
