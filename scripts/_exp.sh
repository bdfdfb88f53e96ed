for v in 4 1 2 3 4; do
  OGS_BLEND_PREFETCH=$v timeout -k 10 120 python bench.py --steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-kmeans > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/exp_$v.json").read().strip().splitlines()[-1])
print($v, d["ms_per_step"], d["value"], d["kernels_ms_per_step"])
PY
done
