# A/B runs of bench.py under different environments / libraries on one box: edit the lines at the bottom
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-extra-workloads --no-cpu-baseline --no-kmeans > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
k = {a: round(b, 4) for a, b in d["kernels_ms_per_step"].items() if "blend_b" in a}
s = {a: round(b, 4) for a, b in d["stage1_pass"]["kernels_ms"].items() if "blend_b" in a}
print(sys.argv[1], round(d["ms_per_step"], 4), "stage1", round(d["stage1_pass"]["ms_per_step"], 4), k, s)
PY
}
timeout -k 10 900 python -m pytest tests/test_10_raster_gpu.py -x -q -m gpu 2>&1 | tail -3 && \
run new && run old OGS_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_in/libogs_prev.so && run new2 && run old2 OGS_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_in/libogs_prev.so
