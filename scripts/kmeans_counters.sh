#!/bin/bash
# SQ counters of the k-means pass (dev helper; run on the GPU box through gpurun):  bash scripts/kmeans_counters.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_km
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS -d $OUT/a -o a -- python3 scripts/kmeans_prof.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM -d $OUT/b -o b -- python3 scripts/kmeans_prof.py > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM -d $OUT/c -o c -- python3 scripts/kmeans_prof.py > $OUT/c.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/d -o d -- python3 scripts/kmeans_prof.py > $OUT/d.log 2>&1 || true
python3 - <<'PY'
import glob, sqlite3, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/prof_km/**/*_results.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    try:
        rows = list(cur.execute("select dispatch_id, kernel_name, grid_size, counter_name, sum(value) from counters_collection group by dispatch_id, kernel_name, counter_name"))
    except Exception as e:
        print(f, e); continue
    for _, k, grid, c, v in rows:
        if "kmeans_mfma_pass" not in k or grid < 2048 * 256:
            continue
        key = "accum" if "true" in k.split("kmeans_mfma_pass_kernel<")[1].split(">")[0] else "assign"
        a = acc[key][c]; a[0] += v; a[1] += 1
out = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open("gpurun_out/prof_km/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
