#!/bin/bash
# A-B of one environment switch of the library on ONE box: the default and `VAR=VALUE` run the same bench command, twice each,
# interleaved.   usage: [WL=workload] bash scripts/ab_env.sh VAR=VALUE
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
WL=${WL:-S1M-1080p}
for rep in 1 2; do
for v in default "$1"; do
  tag=$(echo "$v" | tr '=' '_')
  if [ "$v" = default ]; then PRE=""; else PRE="$v"; fi
  env $PRE timeout -k 10 200 python bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline --no-kmeans --no-extra-workloads --no-extras > gpurun_out/abe_${WL}_${tag}_$rep.json 2> gpurun_out/abe_${WL}_${tag}_$rep.err || { echo FAIL $v; tail -3 gpurun_out/abe_${WL}_${tag}_$rep.err; continue; }
  python -c "
import json
d=json.load(open('gpurun_out/abe_${WL}_${tag}_$rep.json'))
k=d['kernels_ms_per_step']
print('$WL', '$v', 'rep$rep', round(d['ms_per_step'],4), round(d['value'],1), {n:round(x,4) for n,x in list(k.items())[:4]})
"
done
done
