#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory (rocprofv3 CSVs) into a short
text summary for profiles/.

    python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name>
"""
import collections
import csv
import glob
import os
import re
import sys

OURS = ("crossprod", "prep_dense", "colstats", "rowstats", "rowsum", "groupsum",
        "densify", "pbc_", "reduce_partials", "mirror", "transpose_", "scan_tile", "scan_add")


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main(src, dst):
    lines = []
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        lines.append("rocprofv3 --kernel-trace --stats  (top kernels by total time + all svt kernels)")
        lines.append(f"{'kernel':72s} {'calls':>6s} {'avg_us':>12s} {'pct':>7s}")
        for i, r in enumerate(csv.DictReader(open(stats[0]))):
            nm = short(r["Name"])
            if i < 6 or any(k in nm for k in OURS):
                lines.append(f"{nm:72s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:12.1f} "
                             f"{float(r['Percentage']):7.2f}")
    pmc = {}
    for tag, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        fs = glob.glob(os.path.join(src, f"pmc_{tag}", "*", "*_counter_collection.csv"))
        if not fs:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == ctr:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if any(s in k for s in OURS):
                pmc.setdefault(k, {})[ctr] = sum(v) / len(v)
    if pmc:
        lines.append("")
        lines.append("rocprofv3 --pmc (separate passes), per-launch averages, KiB as reported")
        for k, d in pmc.items():
            lines.append(f"{k:72s} " + "  ".join(f"{c}={x:.0f}" for c, x in d.items()))
        lines.append("FETCH_SIZE on gfx950 tallies 128-B requests as 64 B (MI355X_MICROARCH.md, HBM):")
        lines.append("corrected HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.")
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    open(dst + "_summary.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
