# SQ counters of the blend (PROF_RE=kmeans: k-means) kernels under a given environment: bash scripts/prof_sq.sh TAG [VAR=VALUE ...]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-workloads ${PROF_BENCH_FLAGS---no-kmeans}"
rm -rf gpurun_out/sq_$TAG; mkdir -p gpurun_out/sq_$TAG
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM -d gpurun_out/sq_$TAG/a -o s -- $B > gpurun_out/sq_$TAG/a.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d gpurun_out/sq_$TAG/b -o s -- $B > gpurun_out/sq_$TAG/b.log 2>&1 &&
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE -d gpurun_out/sq_$TAG/c -o s -- $B > gpurun_out/sq_$TAG/c.log 2>&1
python3 - $TAG <<'PY'
import sqlite3, glob, collections, json, sys
tag = sys.argv[1]
out = collections.defaultdict(dict)
for d in "abc":
    for f in glob.glob(f"gpurun_out/sq_{tag}/{d}/**/*_results.db", recursive=True):
        cur = sqlite3.connect(f).cursor()
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        try:
            rows = cur.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection group by dispatch_id, kernel_name, counter_name").fetchall()
        except Exception as e:
            print(d, e); continue
        for _, kn, cn, val in rows:
            import re, os
            m = re.search(os.environ.get("PROF_RE", "blend_") + r"\w+(<\w+)?", kn)
            if m:
                k = m.group(0)
                acc[k][cn][0] += val; acc[k][cn][1] += 1
        for k, cs in acc.items():
            for c, v in cs.items():
                out[k][c] = round(v[0] / v[1])
json.dump(out, open(f"gpurun_out/sq_{tag}.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
import shutil
for d in "abc": shutil.rmtree(f"gpurun_out/sq_{tag}/{d}", ignore_errors=True)
PY
