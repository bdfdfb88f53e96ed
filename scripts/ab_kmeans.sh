#!/bin/bash
# A-B of library variants on the k-means leg of bench.py (coarse Lloyd call at N = 2 M, k = 64, d = 9 + the leaf level)
# usage: bash scripts/ab_kmeans.sh variant1 variant2 ...      (variants: python -m opengaussian_amd.build --variant NAME ...)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=$PWD/opengaussian_amd/lib/variants
for rep in 1 2; do
for v in default "$@"; do
  if [ "$v" = default ]; then unset OGS_LIB_PATH; else export OGS_LIB_PATH=$V/libogs_hip_$v.so; fi
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-workloads --no-extras > gpurun_out/abk_${v}_$rep.json 2> gpurun_out/abk_${v}_$rep.err || { echo FAIL $v; tail -3 gpurun_out/abk_${v}_$rep.err; continue; }
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/abk_${v}_$rep.json') if l.startswith('{')][-1])
k=d['kmeans']
print('$v', 'rep$rep', 'it/s', round(k['it_per_s']), 'ms/call', round(k['ms_per_call'],4), {n:round(x,1) for n,x in k['kernels_us_per_launch'].items()})
"
done
done
