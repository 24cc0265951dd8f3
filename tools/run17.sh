cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "A" "TEMP_GEMM_RESIDENT=0"; do
  if [ "$v" = "A" ]; then e=""; else e="$v"; fi
  env $e python bench.py --workload S-hbm --steps 5 --warmup 2 > gpurun_out/b17.json 2> gpurun_out/b17.err
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/b17.json').read().strip().splitlines()[-1])
print(sys.argv[1], round(d['ms_per_step'],3), {k:(round(v['avg_ms'],3)) for k,v in d['kernels'].items()})
PY
done
