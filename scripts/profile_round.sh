#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats            per-kernel durations of the bench command
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE   HBM bytes per launch (separate passes, as the MI355X guide prescribes)
#   3. --pmc SQ_* passes                  VALU / SALU / SMEM instruction counts, wave cycles
# and post-process them into profiles/ (tracked).  Usage: bash scripts/profile_round.sh r02
set -e
TAG=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT profiles
BENCH="python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kmeans --no-extras"
echo "[profile] kernel trace"; rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
echo "[profile] FETCH_SIZE";   rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- $BENCH > $OUT/fetch.log 2>&1
echo "[profile] WRITE_SIZE";   rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- $BENCH > $OUT/write.log 2>&1
echo "[profile] SQ insts";     rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $OUT/sq1 -o s -- $BENCH > $OUT/sq1.log 2>&1
echo "[profile] SQ cycles";    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM -d $OUT/sq2 -o s -- $BENCH > $OUT/sq2.log 2>&1 || true
STATS=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp "$STATS" profiles/${TAG}_kernel_stats.csv
python3 scripts/collect_pmc.py $OUT/fetch $OUT/write profiles/${TAG}_pmc_fetch_write_per_launch.json profiles/pmc_traffic.json
python3 scripts/collect_sq.py profiles/${TAG}_sq_counters_per_launch.json $OUT/sq1 $OUT/sq2
python3 - <<PY
import json, subprocess
from opengaussian_amd import _lib
ver = int(_lib.lib().ogs_version())
src = "scripts/profile_round.sh $TAG: rocprofv3 --pmc passes of '$BENCH' (S1M-1080p fused pass, 8 views cycled)"
t = json.load(open("profiles/pmc_traffic.json")); t["_ogs_version"] = ver; t["_source"] = src + "; bytes = 2*FETCH_SIZE + WRITE_SIZE (KiB -> B), per launch"
json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)
sq = json.load(open("profiles/${TAG}_sq_counters_per_launch.json"))
out = {k: {"SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": v.get("SQ_INSTS_SALU"), "SQ_INSTS_SMEM": v.get("SQ_INSTS_SMEM"), "SQ_WAVES": v.get("SQ_WAVES")} for k, v in sq.items() if "SQ_INSTS_VALU" in v}
out["_ogs_version"] = ver; out["_source"] = src + "; per-launch averages"
json.dump(out, open("profiles/sq_valu.json", "w"), indent=1)
print("profiles written for ogs_version", ver)
PY
head -25 profiles/${TAG}_kernel_stats.csv
