#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats            per-kernel durations of the bench command
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE   HBM bytes per launch (separate passes, as the MI355X guide prescribes)
#   3. --pmc SQ_* passes                  VALU / SALU / SMEM instruction counts, wave cycles (headline workload only)
# and post-process them into profiles/ (tracked).
# usage: bash scripts/profile_round.sh <tag> [workload]       e.g.  r04            (S1M-1080p, the headline: all passes)
#                                                                   r04_c4 C4-2M-648   (an `extras` line: trace + HBM bytes)
set -e
TAG=${1:-r04}
WL=${2:-S1M-1080p}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT profiles
BENCH="python3 bench.py --workload $WL --steps 12 --warmup 3 --no-cpu-baseline --no-kmeans --no-extras --no-extra-workloads"
echo "[profile] $WL kernel trace"; rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
echo "[profile] $WL FETCH_SIZE";   rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- $BENCH > $OUT/fetch.log 2>&1
echo "[profile] $WL WRITE_SIZE";   rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- $BENCH > $OUT/write.log 2>&1
if [ "$WL" = "S1M-1080p" ]; then
  echo "[profile] SQ insts";     rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $OUT/sq1 -o s -- $BENCH > $OUT/sq1.log 2>&1
  echo "[profile] SQ cycles";    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM -d $OUT/sq2 -o s -- $BENCH > $OUT/sq2.log 2>&1 || true
fi
VER=$(python3 -c "from opengaussian_amd import _lib; print(int(_lib.lib().ogs_version()))")
python3 scripts/collect_rocpd.py $OUT $TAG $VER "$BENCH" $WL
# the raw rocpd databases are tens of MB per pass: gpurun merges at most 64 MiB of gpurun_out/ back -- keep the logs and the
# summaries (a copy of what went to profiles/) only
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* profiles/pmc_traffic*.json profiles/sq_valu.json gpurun_out/profiles_out/ 2>/dev/null || true
