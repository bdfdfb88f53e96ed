export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export OGS_BLEND_ROWS_BWD=1
B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kmeans --no-extras --no-extra-workloads"
rm -rf gpurun_out/pr; mkdir -p gpurun_out/pr
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM -d gpurun_out/pr/a -o s -- $B > gpurun_out/pr/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d gpurun_out/pr/b -o s -- $B > gpurun_out/pr/b.log 2>&1
python3 - <<'PY'
import sqlite3, glob, collections
for d in ("gpurun_out/pr/a", "gpurun_out/pr/b"):
    for f in glob.glob(d + "/**/*_results.db", recursive=True):
        cur = sqlite3.connect(f).cursor()
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for _, kn, cn, val in cur.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection group by dispatch_id, kernel_name, counter_name"):
            if "blend_backward" in kn or "blend_forward" in kn:
                k = "bwd_rows" if "backward_rows" in kn else "bwd_quad" if "blend_backward" in kn else "fwd_rows" if "forward_rows" in kn else "fwd_quad"
                acc[k][cn][0] += val; acc[k][cn][1] += 1
        for k, cs in acc.items():
            print(k, {c: round(v[0] / v[1]) for c, v in cs.items()})
PY
