# A/B runs of bench.py under different environments on one box: edit the `run` lines
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-extra-workloads --no-cpu-baseline --no-kmeans > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
k = {a: round(b, 4) for a, b in d["kernels_ms_per_step"].items()}
print(sys.argv[1], round(d["ms_per_step"], 4), "stage1", round(d["stage1_pass"]["ms_per_step"], 4), k)
PY
}
run cull && run full OGS_FULL_BINNING=1 && run cull2 && run full2 OGS_FULL_BINNING=1
