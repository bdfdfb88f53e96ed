# A/B runs of bench.py under different environments / libraries on one box: edit the lines at the bottom
cd $GRAFT_REPO_ROOT
c2() { env "$@" timeout -k 10 300 python bench.py --workload C2-100k-800 --steps 300 --warmup 20 --no-extra-workloads --no-cpu-baseline --no-kmeans 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', sys.argv[1:], d['ms_per_step'])" "$@"; }
timeout -k 10 600 python -m pytest tests/test_10_raster_gpu.py -x -q -m gpu -k "forward_parity or deferred or grouped or tiny" 2>&1 | tail -2 && c2 A=1 && c2 A=2 && c2 A=3 && timeout -k 10 200 python scripts/host_overhead.py 2>/dev/null | tail -40 | head -60
