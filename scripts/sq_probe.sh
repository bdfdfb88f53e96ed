#!/bin/bash
# Ad-hoc SQ counter probe of the bench command (two --pmc passes); prints per-launch averages for the blend kernels.
# usage (through gpurun, from the repo root): bash scripts/sq_probe.sh [tag]
set -e
TAG=${1:-probe}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/sq_$TAG
rm -rf $OUT && mkdir -p $OUT
BENCH="python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kmeans --no-extras"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_MFMA SQ_BUSY_CYCLES -d $OUT/a -o s -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/b -o s -- $BENCH > $OUT/b.log 2>&1
python3 - <<PY
import json, sys
sys.path.insert(0, "scripts")
from collect_rocpd import counters
c = counters(["$OUT/a", "$OUT/b"])
out = {k: v for k, v in c.items() if "blend" in k}
json.dump(out, open("$OUT/blend_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
