"""Post-process rocprofv3 --pmc passes of bench.py into per-kernel, per-launch averages of every counter found.

usage: python scripts/collect_sq.py <out_json> <dir> [<dir> ...]"""
import collections
import csv
import glob
import json
import re
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "ogs::" not in r["Kernel_Name"]:
                    continue
                m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
                if not m:
                    continue
                name = re.sub(r"<(\d+),[^>]*>", r"<\1>", m.group(1))
                a = acc[name][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    table = {k: {c: v[0] / max(v[1], 1) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
    for k, cs in table.items():
        if "SQ_WAVE_CYCLES" in cs and "SQ_ACTIVE_INST_VALU" in cs:
            # SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count in units of 4 cycles per wave-quad on gfx9-family SQs: keep ratios only
            cs["valu_active_over_wave_cycles"] = cs["SQ_ACTIVE_INST_VALU"] / cs["SQ_WAVE_CYCLES"]
        if "SQ_WAVE_CYCLES" in cs and "SQ_WAIT_INST_ANY" in cs:
            cs["wait_inst_over_wave_cycles"] = cs["SQ_WAIT_INST_ANY"] / cs["SQ_WAVE_CYCLES"]
        if "SQ_INSTS_VALU" in cs and "SQ_WAVES" in cs:
            cs["valu_insts_per_wave"] = cs["SQ_INSTS_VALU"] / cs["SQ_WAVES"]
    json.dump(table, open(out, "w"), indent=1)
    for k in ("blend_backward_kernel<9>", "blend_forward_kernel<9>", "pack_sorted_kernel<9>"):
        if k in table:
            print(k, {c: (round(v, 3) if v < 10 else round(v)) for c, v in table[k].items()})


if __name__ == "__main__":
    main()
