cd $GRAFT_REPO_ROOT
c2() { env "$@" timeout -k 10 300 python bench.py --workload C2-100k-800 --steps 300 --warmup 20 --no-extra-workloads --no-cpu-baseline --no-kmeans 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', sys.argv[1:], d['ms_per_step'])" "$@"; }
c2 A=mt && c2 OGS_BENCH_AUTOGRAD_MT=0 && c2 A=mt && c2 OGS_BENCH_AUTOGRAD_MT=0
