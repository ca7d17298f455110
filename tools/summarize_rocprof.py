#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel_stats / counter_collection) into a short
text summary that is committed under profiles/.

    python tools/summarize_rocprof.py gpurun_out/r1_prof > profiles/r01_kernel_stats.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=110):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main(root):
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
        print(f"# {os.path.relpath(f, root)}  (rocprofv3 --kernel-trace --stats)")
        print(f"{'calls':>6} {'avg_us':>12} {'min_us':>12} {'max_us':>12} {'pct':>6}  kernel")
        for row in csv.DictReader(open(f)):
            print(f"{row['Calls']:>6} {float(row['AverageNs'])/1e3:12.1f} {float(row['MinNs'])/1e3:12.1f} "
                  f"{float(row['MaxNs'])/1e3:12.1f} {float(row['Percentage']):6.2f}  {short(row['Name'])}")
        print()
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        print(f"# {os.path.relpath(f, root)}  (rocprofv3 --pmc)")
        acc = defaultdict(lambda: defaultdict(list))
        meta = {}
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"], 90)
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[k] = (row.get("VGPR_Count", "?"), row.get("SGPR_Count", "?"), row.get("LDS_Block_Size", "?"),
                       row.get("Grid_Size", "?"), row.get("Workgroup_Size", "?"))
        for k, counters in acc.items():
            if "epsm" not in k and "gather128" not in k:
                continue
            v = meta[k]
            print(f"kernel {k}\n  vgpr={v[0]} sgpr={v[1]} lds={v[2]} grid={v[3]} wg={v[4]}")
            for c, vals in sorted(counters.items()):
                print(f"  {c:28s} dispatches={len(vals):3d} mean={sum(vals)/len(vals):.6g} min={min(vals):.6g} max={max(vals):.6g}")
        print()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
