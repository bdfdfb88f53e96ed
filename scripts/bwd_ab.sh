# A/B runs of bench.py under different environments / libraries on one box: edit the lines at the bottom
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-extra-workloads --no-cpu-baseline --no-kmeans > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
k = {a: round(b, 4) for a, b in d["kernels_ms_per_step"].items() if "radix" in a and "16" in a or "pack" in a}
print(sys.argv[1], round(d["ms_per_step"], 4), "stage1", round(d["stage1_pass"]["ms_per_step"], 4), k)
PY
}
run b7 && run b5 OGS_TILE_SORT_FIRST_BITS=5 && run b6 OGS_TILE_SORT_FIRST_BITS=6 && run b8 OGS_TILE_SORT_FIRST_BITS=8 && run b7b
