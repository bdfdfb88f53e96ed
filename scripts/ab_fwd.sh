#!/bin/bash
# A-B of forward-kernel variant builds (full per-kernel breakdown pass: the forward is not the bracketed dominant kernel)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=$PWD/opengaussian_amd/lib/variants
WL=${WL:-S1M-1080p}
for rep in 1 2; do
for v in default "$@"; do
  if [ "$v" = default ]; then unset OGS_LIB_PATH; else export OGS_LIB_PATH=$V/libogs_hip_$v.so; fi
  timeout -k 10 200 python bench.py --workload $WL --steps 60 --warmup 10 --no-cpu-baseline --no-kmeans --no-extra-workloads > gpurun_out/abf_${v}_$rep.json 2> gpurun_out/abf_${v}_$rep.err || { echo FAIL $v; continue; }
  python -c "
import json
d=json.load(open('gpurun_out/abf_${v}_$rep.json'))
print('$WL', '$v', 'rep$rep', round(d['ms_per_step'],4), {n:round(x,4) for n,x in list(d['kernels_ms_per_step'].items())[:3]}, 'stage1', round(d['stage1_pass']['ms_per_step'],4))
"
done
done
