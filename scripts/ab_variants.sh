#!/bin/bash
# A-B timing of library variants (opengaussian_amd/build.py --variant NAME ...) on ONE box: the default library and every
# lib/variants/libogs_hip_<name>.so named on the command line run the same bench command, twice each, interleaved.
# usage: bash scripts/ab_variants.sh [-w WORKLOAD] name1 name2 ...      (results: gpurun_out/ab_<workload>_<name>_<rep>.json)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
WL=S1M-1080p
if [ "$1" = "-w" ]; then WL=$2; shift 2; fi
for rep in 1 2; do
  for v in default "$@"; do
    if [ "$v" = default ]; then unset OGS_LIB_PATH; else export OGS_LIB_PATH=$PWD/opengaussian_amd/lib/variants/libogs_hip_$v.so; fi
    timeout -k 10 200 python bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline --no-kmeans --no-extra-workloads --no-extras \
        > gpurun_out/ab_${WL}_${v}_$rep.json 2> gpurun_out/ab_${WL}_${v}_$rep.err || { echo "FAILED $v"; tail -5 gpurun_out/ab_${WL}_${v}_$rep.err; exit 1; }
    python - "$WL" "$v" "$rep" <<'PY'
import json, sys
wl, v, rep = sys.argv[1:4]
d = json.load(open(f"gpurun_out/ab_{wl}_{v}_{rep}.json"))
k = d["kernels_ms_per_step"]
top = {n.split("_kernel")[0]: round(x, 4) for n, x in sorted(k.items(), key=lambda kv: -kv[1])[:4]}
print(f"{wl:12s} {v:10s} rep{rep} ms/step {d['ms_per_step']:.4f}  {top}", flush=True)
PY
  done
done
