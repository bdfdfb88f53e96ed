#!/bin/bash
# A-B of library variant builds with the full per-kernel breakdown pass (for kernels that are not the bracketed dominant one)
# usage: [WL=workload] bash scripts/ab_fwd.sh variant1 variant2 ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=$PWD/opengaussian_amd/lib/variants
WL=${WL:-S1M-1080p}
for rep in 1 2; do
for v in default "$@"; do
  if [ "$v" = default ]; then unset OGS_LIB_PATH; else export OGS_LIB_PATH=$V/libogs_hip_$v.so; fi
  timeout -k 10 200 python bench.py --workload $WL --steps 60 --warmup 10 --no-cpu-baseline --no-kmeans --no-extra-workloads > gpurun_out/abf_${WL}_${v}_$rep.json 2> gpurun_out/abf_${WL}_${v}_$rep.err || { echo FAIL $v; continue; }
  python -c "
import json
d=json.load(open('gpurun_out/abf_${WL}_${v}_$rep.json'))
k=d['kernels_ms_per_step']
print('$WL', '$v', 'rep$rep', round(d['ms_per_step'],4), {n:round(x,4) for n,x in list(k.items())[:5]}, 'sorts', round(sum(x for n,x in k.items() if 'radix' in n),4), 'stage1', round(d['stage1_pass']['ms_per_step'],4))
"
done
done
