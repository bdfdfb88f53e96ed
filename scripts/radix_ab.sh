#!/bin/bash
# A-B of the radix sort variants on ONE box (OGS_RADIX=legacy: three launches per pass; default: one launch per pass with
# decoupled look-back).  Prints ms/step and the radix kernels' ms per step.  (The tile-size switch OGS_SWEEP_ITEMS of
# round 3 is gone: forced tile sizes are an argument of the radix self-test hook, tests/test_13_radix_gpu.py.)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for wl in C2-100k-800 S1M-1080p; do
  for mode in legacy sweep; do
    case $mode in
      legacy) export OGS_RADIX=legacy;;
      sweep) unset OGS_RADIX;;
    esac
    timeout -k 10 200 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-kmeans > gpurun_out/ab_${wl}_${mode}.json 2> gpurun_out/ab_${wl}_${mode}.err || exit 1
    python - "$wl" "$mode" <<'PY'
import json, sys
wl, mode = sys.argv[1:3]
d = json.load(open(f"gpurun_out/ab_{wl}_{mode}.json"))
k = d["kernels_ms_per_step"]
rad = {n: round(v, 4) for n, v in k.items() if "radix" in n}
print(f"{wl:12s} {mode:8s} ms/step {d['ms_per_step']:.4f} kernel-sum {sum(k.values()):.4f} radix-sum {sum(rad.values()):.4f} {rad}")
PY
  done
done
