timeout -k 10 600 python -m pytest tests/test_10_raster_gpu.py tests/test_60_integration_gpu.py -x -q > gpurun_out/fo_t.log 2>&1; echo "exit=$?" >> gpurun_out/fo_t.log; tail -3 gpurun_out/fo_t.log
for v in A new A new; do
  if [ $v = A ]; then export OGS_LIB_PATH=$PWD/opengaussian_amd/lib/ab/A.so; else unset OGS_LIB_PATH; fi
  timeout -k 10 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kmeans > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["ms_per_step"],4), d["stage1_pass"])
PY
done
