# timing experiments on the backward blend (results are wrong with OGS_BLEND_PREFETCH bits >= 8 set): bash scripts/bwd_ab.sh
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-extra-workloads --no-cpu-baseline --no-kmeans > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
k = {a: round(b, 4) for a, b in d["kernels_ms_per_step"].items() if "blend_b" in a}
print(sys.argv[1], round(d["ms_per_step"], 4), k)
PY
}
OGS_BLEND_BWD_VREC=2 timeout -k 10 600 python -m pytest tests/test_10_raster_gpu.py tests/test_11_render_gpu.py -x -q -m gpu 2>&1 | tail -3 && \
run quad && run vrec32 OGS_BLEND_BWD_VREC=2 && run vrec64 OGS_BLEND_BWD_VREC=1
