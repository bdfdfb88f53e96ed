"""GPU timeline of a `rocprofv3 --kernel-trace` run of bench.py (rocpd / SQLite): per step, kernel-busy time, idle gaps
between consecutive dispatches and which kernels they follow.  usage: python scripts/timeline_gaps.py <dir with *_results.db> [steps]"""
import collections
import glob
import json
import re
import os
import sqlite3
import sys


def short(n):
    m = re.search(r"(\w+_kernel\w*|__amd_rocclr_\w+|at::native::\w+)", n)
    return m.group(1) if m else n[:40]


def main():
    d = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    f = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
    cur = sqlite3.connect(f).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    # timed steps: the last `nsteps` occurrences of the backward blend mark the step ends
    ends = [i for i, r in enumerate(rows) if "preprocess_backward_kernel" in r[0]]
    ends = ends[-(nsteps + 1):]
    lo, hi = ends[0] + 1, ends[-1] + 1
    seg = rows[lo:hi]
    span = seg[-1][2] - seg[0][1]
    busy = sum(e - s for _, s, e in seg)
    gaps = collections.defaultdict(lambda: [0, 0])
    for (n0, s0, e0), (n1, s1, e1) in zip(seg, seg[1:]):
        g = max(s1 - e0, 0)
        k = (short(n0), short(n1))
        gaps[k][0] += g
        gaps[k][1] += 1
    per = collections.defaultdict(lambda: [0, 0])
    for n, s0, e0 in seg:
        per[short(n)][0] += e0 - s0
        per[short(n)][1] += 1
    out = {"kernel_us_per_step": {k: round(v[0] / nsteps / 1e3, 2) for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])},
           "steps": nsteps, "dispatches_per_step": len(seg) / nsteps, "span_ms_per_step": span / nsteps / 1e6,
           "busy_ms_per_step": busy / nsteps / 1e6, "idle_ms_per_step": (span - busy) / nsteps / 1e6,
           "largest_gaps_us_per_step": [
               {"after": k[0], "before": k[1], "us_per_step": v[0] / nsteps / 1e3, "count_per_step": v[1] / nsteps}
               for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
