#!/bin/bash
# SQ counters + kernel trace of the stage-1 step (dev helper, GPU box)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_s1
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS -d $OUT/a -o a -- python3 scripts/stage1_prof.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM -d $OUT/b -o b -- python3 scripts/stage1_prof.py > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d $OUT/c -o c -- python3 scripts/stage1_prof.py > $OUT/c.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE -d $OUT/f -o f -- python3 scripts/stage1_prof.py > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/w -o w -- python3 scripts/stage1_prof.py > $OUT/w.log 2>&1
python3 - <<'PY'
import glob, sqlite3, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/prof_s1/**/*_results.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    for _, k, c, v in cur.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection group by dispatch_id, kernel_name, counter_name"):
        m = re.search(r"(blend_\w+_kernel)<", k)
        if not m: continue
        a = acc[m.group(1)][c]; a[0] += v; a[1] += 1
out = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open("gpurun_out/prof_s1/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
