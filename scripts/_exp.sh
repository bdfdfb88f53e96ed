export TMPDIR=/tmp
rm -rf gpurun_out/tl && mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace -d gpurun_out/tl -o t -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kmeans --no-extras > gpurun_out/tl/log.txt 2>&1
python3 scripts/timeline_gaps.py gpurun_out/tl 12 > gpurun_out/tl/gaps.json; cat gpurun_out/tl/gaps.json
