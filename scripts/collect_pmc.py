"""Post-process two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) of bench.py into per-kernel HBM bytes per launch:

    traffic = 2 * FETCH_SIZE (gfx950 reports half of wide streaming reads) + WRITE_SIZE      [KiB -> bytes]

usage: python scripts/collect_pmc.py <fetch_dir> <write_dir> <out_json> [<pmc_traffic_json>]"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(dirname, counter):
    files = glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
            if not m or "ogs::" not in r["Kernel_Name"]:
                continue
            name = m.group(1)
            name = re.sub(r"<(\d+),[^>]*>", r"<\1>", name)
            a = acc[name]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: v[0] / max(v[1], 1) for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, nf = per_kernel(fdir, "FETCH_SIZE")
    write, nw = per_kernel(wdir, "WRITE_SIZE")
    table, traffic = {}, {}
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
        b = (2.0 * fk + wk) * 1024.0
        table[k] = {"FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk, "launches_sampled": nf.get(k, 0),
                    "hbm_bytes_corrected_per_launch": b}
        traffic[k] = b
    json.dump(table, open(out, "w"), indent=1)
    if len(sys.argv) > 4:
        json.dump(traffic, open(sys.argv[4], "w"), indent=1)
    for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]):
        print(f"{k:40s} {v / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main()
