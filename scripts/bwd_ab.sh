# A/B runs of bench.py under different environments / libraries on one box: edit the lines at the bottom
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-extra-workloads --no-cpu-baseline --no-kmeans > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
print(sys.argv[1], round(d["ms_per_step"], 4), "stage1", round(d["stage1_pass"]["ms_per_step"], 4))
PY
}
c2() { env "$@" timeout -k 10 300 python bench.py --workload C2-100k-800 --steps 300 --warmup 20 --no-extra-workloads --no-cpu-baseline --no-kmeans 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', sys.argv[1:], d['ms_per_step'])" "$@"; }
timeout -k 10 900 python -m pytest tests/test_10_raster_gpu.py tests/test_11_render_gpu.py -x -q -m gpu 2>&1 | tail -3 && run new && c2 A=new && run new2 && c2 A=new
