"""Post-process the rocprofv3 (rocpd / SQLite) outputs of scripts/profile_round.sh into the tracked summaries under
profiles/:

    <tag>_kernel_stats.csv                 --kernel-trace --stats pass: calls, total / average duration per kernel
    <tag>_pmc_fetch_write_per_launch.json  FETCH_SIZE / WRITE_SIZE passes (separate runs, as the MI355X guide prescribes);
                                           HBM bytes per launch = 2 * FETCH_SIZE (gfx950 reports half of wide streaming
                                           reads) + WRITE_SIZE   [KiB -> bytes]
    <tag>_sq_counters_per_launch.json      SQ_* passes: per-launch averages of every counter
    pmc_traffic.json, sq_valu.json         what bench.py attaches to its JSON line (tagged with ogs_version); for a workload other
                                           than the headline: pmc_traffic_<workload>.json (the `extras` lines)

usage: python scripts/collect_rocpd.py <prof_dir> <tag> <ogs_version> "<bench command>" [workload] """
import collections
import csv
import glob
import json
import os
import re
import sqlite3
import sys


def short(name):
    if "ogs::" not in name:
        return None
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
    if not m:
        return None
    return re.sub(r"<(\d+)[^>]*>", r"<\1>", m.group(1))          # blend_backward_kernel<9, 3, false, double> -> <9>


# dominant read pattern of every kernel of the step (second field: is it the pattern the guide's x2 is calibrated for?)
READ_PATTERN = {
    "preprocess_kernel": ("coalesced vector loads of the per-Gaussian inputs (SH rows 192 B per lane as 16-byte loads)", True),
    "preprocess_backward_kernel": ("coalesced 16-byte vector loads of inputs + the 128-byte fp64 gradient record", True),
    "pack_sorted_kernel": ("4-byte point-list stream + 32 / 48-byte record gathers by sorted id (16-byte loads)", True),
    "duplicate_kernel": ("4-byte / 16-byte coalesced loads of the depth-ordered per-Gaussian state", True),
    "radix_hist_kernel": ("4-byte-per-lane coalesced key stream", False),
    "radix_scatter_kernel": ("4-byte-per-lane coalesced key / value streams", False),
    "radix_onesweep_kernel": ("4-byte-per-lane coalesced key / value streams + status-word polls", False),
    "radix_hist_all_kernel": ("4-byte-per-lane coalesced key stream", False),
    "radix_rowscan_kernel": ("small table, L2 resident", False),
    "scan_reduce_kernel": ("4-byte-per-lane loads through a gather index", False),
    "scan_apply_kernel": ("4-byte-per-lane loads through a gather index", False),
    "tile_ranges_kernel": ("4-byte-per-lane coalesced key stream", False),
    "tile_order_kernel": ("small table", False),
    "blend_forward_kernel": ("SCALAR loads of the record stream (s_load_dwordx16) + one-dword-per-line vector prefetch touches", False),
    "blend_forward_rows_kernel": ("per-lane 16-byte gathers of 80-byte records (L2 resident after the line-touch prefetch) + the prefetch touches", False),
    "blend_backward_kernel": ("SCALAR loads of the record stream + line touches + per-pixel 4-byte loads; writes are fp64 atomics", False),
    "blend_backward_feat_kernel": ("SCALAR loads of the record stream + line touches; writes are fp64 atomics", False),
    "blend_backward_feat_lds_kernel": ("per-lane 16-byte gathers of the records' geometry halves + line touches; writes are fp64 atomics", False),
    "pack_blend_chunked_kernel": ("pack's 4-byte point-list stream + 32 / 48-byte record gathers by sorted id (per-lane, random); the records are "
                                  "blended from LDS, the compacted copy and the index streams are written once for the backward", True),
    "pack_blend_forward_kernel": ("pack's 4-byte point-list stream + 32 / 48-byte record gathers by sorted id, then per-lane 16-byte gathers "
                                  "of the records the workgroup just wrote (L2 / L1 resident)", True),
}


def dbs(d):
    return glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)


def kernel_stats(d, out_csv):
    rows = []
    for f in dbs(d):
        cur = sqlite3.connect(f).cursor()
        rows += list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    rows.sort(key=lambda r: -r[2])
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow(r)
    return rows


def counters(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in dbs(d):
            cur = sqlite3.connect(f).cursor()
            # one row per (dispatch, counter, [dimension instance]): sum the instances of a dispatch first
            q = "select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection group by dispatch_id, kernel_name, counter_name"
            for _, kname, cname, val in cur.execute(q):
                k = short(kname)
                if k is None:
                    continue
                a = acc[k][cname]
                a[0] += float(val); a[1] += 1
    return {k: {c: v[0] / max(v[1], 1) for c, v in sorted(cs.items())} | {"_launches": max(v[1] for v in cs.values())}
            for k, cs in sorted(acc.items())}


def main():
    d, tag, ver, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    wl = sys.argv[5] if len(sys.argv) > 5 else "S1M-1080p"
    headline = wl == "S1M-1080p"
    os.makedirs("profiles", exist_ok=True)
    rows = kernel_stats(os.path.join(d, "trace"), f"profiles/{tag}_kernel_stats.csv")
    fetch = counters([os.path.join(d, "fetch")])
    write = counters([os.path.join(d, "write")])
    table, traffic, how = {}, {}, {}
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k, {}).get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
        # The guide's x2 on FETCH_SIZE is calibrated for wide (16 B per lane) coalesced streaming reads only.  Per kernel:
        # which read pattern dominates, hence which figure is quoted; both bounds are always kept.
        pattern, streaming = READ_PATTERN.get(k.split("<")[0], ("unclassified", False))
        lo, hi = (fk + wk) * 1024.0, (2.0 * fk + wk) * 1024.0
        b = hi
        table[k] = {"FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk,
                    "launches_sampled": fetch.get(k, {}).get("_launches", 0), "read_pattern": pattern,
                    "fetch_correction": "x2 (calibrated: wide coalesced streaming reads)" if streaming else
                                        "x2 quoted as an UPPER bound (uncalibrated pattern: the true figure lies between the two)",
                    "hbm_bytes_lower_bound_per_launch": lo, "hbm_bytes_corrected_per_launch": b}
        traffic[k] = b
        how[k] = table[k]["fetch_correction"] + " -- " + pattern
    json.dump(table, open(f"profiles/{tag}_pmc_fetch_write_per_launch.json", "w"), indent=1)
    src = f"scripts/profile_round.sh {tag}: rocprofv3 --pmc passes of '{cmd}' ({wl} fused pass, 8 views cycled)"
    traffic["_ogs_version"] = ver
    traffic["_source"] = src + "; bytes = 2*FETCH_SIZE + WRITE_SIZE (KiB -> B), per launch"
    traffic["_correction_per_kernel"] = how
    json.dump(traffic, open("profiles/pmc_traffic.json" if headline else f"profiles/pmc_traffic_{wl}.json", "w"), indent=1)
    if not headline:
        print("kernel stats (top 6):")
        for r in rows[:6]:
            print("  %-60s calls %4d avg %9.1f us  HBM MB/launch %.1f" % ((short(r[0]) or r[0][:60]), r[1], r[3],
                                                                         traffic.get(short(r[0]) or "", 0) / 1e6))
        return
    sq = counters([os.path.join(d, "sq1"), os.path.join(d, "sq2")])
    for k, cs in sq.items():
        if "SQ_INSTS_VALU" in cs and "SQ_WAVES" in cs and cs["SQ_WAVES"]:
            cs["valu_insts_per_wave"] = cs["SQ_INSTS_VALU"] / cs["SQ_WAVES"]
        if "SQ_WAVE_CYCLES" in cs and cs.get("SQ_WAVE_CYCLES"):
            if "SQ_ACTIVE_INST_VALU" in cs:
                cs["valu_active_over_wave_cycles"] = cs["SQ_ACTIVE_INST_VALU"] / cs["SQ_WAVE_CYCLES"]
            if "SQ_WAIT_INST_ANY" in cs:
                cs["wait_inst_over_wave_cycles"] = cs["SQ_WAIT_INST_ANY"] / cs["SQ_WAVE_CYCLES"]
    json.dump(sq, open(f"profiles/{tag}_sq_counters_per_launch.json", "w"), indent=1)
    valu = {k: {c: cs.get(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_WAVES")} for k, cs in sq.items()
            if "SQ_INSTS_VALU" in cs}
    valu["_ogs_version"] = ver
    valu["_source"] = src + "; per-launch averages"
    json.dump(valu, open("profiles/sq_valu.json", "w"), indent=1)
    print("kernel stats (top 6):")
    for r in rows[:6]:
        print("  %-60s calls %4d avg %9.1f us" % ((short(r[0]) or r[0][:60]), r[1], r[3]))
    for k in ("blend_backward_kernel<9>", "pack_blend_chunked_kernel<9>"):
        print(k, "HBM MB/launch", round(traffic.get(k, 0) / 1e6, 1), {c: round(v) for c, v in sq.get(k, {}).items() if c.startswith("SQ_INSTS")})


if __name__ == "__main__":
    main()
