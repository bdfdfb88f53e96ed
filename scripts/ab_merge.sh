#!/bin/bash
# A-B of the backward blend's record merge (round 4): default (windows of OGS_MERGE_W records in a ring of OGS_MERGE_K buffers),
# variant builds with other window shapes, and OGS_BLEND_MERGE=0 (one atomic record per (entry, quadrant)); then the sensitivity
# probes of the chunked forward (full per-kernel breakdown: the forward is not the bracketed dominant kernel).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { # name extra-bench-args env...
  name=$1; shift; extra=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kmeans --no-extra-workloads $extra > gpurun_out/ab2_$name.json 2> gpurun_out/ab2_$name.err || { echo FAIL $name; tail -3 gpurun_out/ab2_$name.err; return; }
  python -c "
import json
d=json.load(open('gpurun_out/ab2_$name.json'))
print('$name', round(d['ms_per_step'],4), {n:round(v,4) for n,v in list(d['kernels_ms_per_step'].items())[:3]})
"
}
V=$PWD/opengaussian_amd/lib/variants
for rep in 1 2; do
run merge_w16k2_$rep --no-extras A=1
run unmerged_$rep --no-extras OGS_BLEND_MERGE=0
run merge_w8k4_$rep --no-extras OGS_LIB_PATH=$V/libogs_hip_w8k4.so
run merge_w16k4_$rep --no-extras OGS_LIB_PATH=$V/libogs_hip_w16k4.so
run merge_w32k2_$rep --no-extras OGS_LIB_PATH=$V/libogs_hip_w32k2.so
done
run fwd_default "" A=1
run fwd_probe1_valu "" OGS_LIB_PATH=$V/libogs_hip_fprobe1.so
run fwd_probe2_lds "" OGS_LIB_PATH=$V/libogs_hip_fprobe2.so
run fwd_probe3_nowriteout "" OGS_LIB_PATH=$V/libogs_hip_fprobe3.so
