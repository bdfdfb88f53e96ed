timeout -k 10 200 python -m cProfile -o gpurun_out/c2.prof bench.py --workload C2-100k-800 --steps 300 --warmup 10 --no-extras --no-cpu-baseline --no-kmeans > gpurun_out/c2_prof.json 2> gpurun_out/c2_prof.err
python - <<PY
import pstats
p=pstats.Stats("gpurun_out/c2.prof"); p.sort_stats("tottime").print_stats(28)
PY
