timeout -k 10 600 python -m pytest tests/test_10_raster_gpu.py tests/test_11_render_gpu.py -x -q 2>&1 | tail -12
for v in A new A new; do
  if [ $v = A ]; then export OGS_LIB_PATH=$PWD/opengaussian_amd/lib/ab/A.so; else unset OGS_LIB_PATH; fi
  timeout -k 10 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kmeans > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
k=d["kernels_ms_per_step"]
print("$v", round(d["ms_per_step"],4), {n:round(k[n],4) for n in k if "pack" in n or "dup" in n or "forward" in n})
PY
done
