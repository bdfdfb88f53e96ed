# kernel timeline of one step: which kernels overlap in time (rocprofv3 --kernel-trace): bash scripts/trace_overlap.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr; mkdir -p gpurun_out/tr
rocprofv3 --kernel-trace -d gpurun_out/tr -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kmeans --no-extras --no-extra-workloads $BENCH_ARGS > gpurun_out/tr/log 2>&1
python3 - <<'PY'
import sqlite3, glob, re
f = glob.glob("gpurun_out/tr/**/*_results.db", recursive=True)[0]
con = sqlite3.connect(f)
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
print(cols)
q = f"select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
try:
    rows = cur.execute(q).fetchall()
except Exception as e:
    print("query failed", e); rows = []
# last full step: find the last preprocess_geom / preprocess_kernel
idx = [i for i, r in enumerate(rows) if "preprocess_geom" in r[0] or re.search(r"preprocess_kernel", r[0])]
if idx:
    i0 = idx[-1]
    t0 = rows[i0][1]
    for r in rows[i0:i0 + 40]:
        nm = re.search(r"(\w+_kernel)", r[0])
        print("%-34s start %8.1f us  dur %7.1f us  queue %s stream %s" % (nm.group(1) if nm else r[0][:30], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4]))
PY
rm -rf gpurun_out/tr/*/ 2>/dev/null; find gpurun_out/tr -name "*.db" -delete
