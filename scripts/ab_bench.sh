# A-B timing of bench.py on ONE box (run through gpurun): every `run` / `c2` line is one child process of bench.py under the given
# environment (OGS_* switches of README.md, or OGS_LIB_PATH=<another build of libogs_hip.so placed under gpurun_in/>).
#   bash scripts/ab_bench.sh > gpurun_out/ab.log
cd $GRAFT_REPO_ROOT
run() {  # name, VAR=VALUE ...: the S1M headline step + the stage-1 step, blend kernels listed
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-extra-workloads --no-cpu-baseline --no-kmeans > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1
  python - $name <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
k = {a: round(b, 4) for a, b in d["kernels_ms_per_step"].items() if "blend" in a}
s = {a: round(b, 4) for a, b in d["stage1_pass"]["kernels_ms"].items() if "blend_b" in a}
print(sys.argv[1], round(d["ms_per_step"], 4), "stage1", round(d["stage1_pass"]["ms_per_step"], 4), k, s)
PY
}
c2() {   # VAR=VALUE ...: the host-paced 100 k-Gaussian workload
  env "$@" timeout -k 10 300 python bench.py --workload C2-100k-800 --steps 300 --warmup 20 --no-extra-workloads --no-cpu-baseline --no-kmeans 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', sys.argv[1:], d['ms_per_step'])" "$@"
}
# A-B-A-B: the pool's boxes differ by a few per cent, a pair of runs on the same box does not
run default && run quadrant_forward OGS_BLEND_ROWS=0 && run default2 && run quadrant_forward2 OGS_BLEND_ROWS=0
