"""Per-bounce kernel times of the wavefront tracer from a rocprofv3 --kernel-trace CSV (dispatches in start order)."""
import csv, glob, sys
from collections import defaultdict
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    seq = defaultdict(list); idx = defaultdict(int)
    for r in rows:
        n = r["Kernel_Name"]
        if "epsm" not in n: continue
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1]
        if "generate" in short:
            idx.clear()
        k = idx[short]; idx[short] += 1
        seq[(short, k)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# " + f)
    for (short, k), v in seq.items():
        print(f"{short:32s} call {k}: n={len(v):3d} avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}")
