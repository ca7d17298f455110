#!/usr/bin/env python3
"""profiles/traffic.json from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as the
MI355X guide prescribes): HBM bytes per launch of the fused backward kernel = 2 x FETCH_SIZE (gfx950 tallies the 128-B
requests of wide streaming reads at 64 B: MI355X_MICROARCH.md "HBM"; re-checked here on epsm_tangent_kernel, whose reads
are known) + WRITE_SIZE (exact for stores and float atomics).  Every entry is stamped with the fingerprint of the kernel
sources it was measured on (bench.kernel_source_hash); bench.py refuses entries of other sources.

    python tools/make_traffic_json.py FETCH_DIR WRITE_DIR [CALIB_FETCH_DIR] --paths N --K 5 --variant manifold --profile bathroom --tag r02_b
    [--packed --gather-calib DIR]   the run was on the native packed log (kernel <..., true>); DIR = FETCH_SIZE of
                                    tools/micro/gather128 (N x 128 B read by per-lane 16-byte gathers), which fixes the
                                    factor for THAT access pattern instead of the x2 of wide coalesced reads
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def counters(root, name):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def pick(acc, needle):
    for k, v in acc.items():
        if needle in k:
            return k, v
    return None, []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir"); ap.add_argument("calib_dir", nargs="?")
    ap.add_argument("--paths", type=int, default=1 << 24); ap.add_argument("--K", type=int, default=5)
    ap.add_argument("--variant", default="manifold"); ap.add_argument("--profile", default="bathroom")
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--packed", action="store_true"); ap.add_argument("--gather-calib", default=None)
    ap.add_argument("--sq-dir", default=None, help="rocprofv3 --pmc run with SQ_INSTS_VALU: stored as sq_insts_valu (bench.py: valu_floor_ms)")
    ap.add_argument("--rdreq-dir", default=None, help="rocprofv3 --pmc run with TCC_EA0_RDREQ_{32B,64B,128B}_sum: the read traffic is then "
                    "the requests by size -- no factor, no calibration (FETCH_SIZE tallies every request at 64 B)")
    a = ap.parse_args()
    import bench
    kern = "epsm_backward_cp_kernel"                                        # <VARIANT, DMODE, PACKED, FLOAT_ROWS>
    needle = ", true, " if a.packed else kern
    fk, fv = pick({k: v for k, v in counters(a.fetch_dir, "FETCH_SIZE").items() if kern in k}, needle)
    wk, wv = pick({k: v for k, v in counters(a.write_dir, "WRITE_SIZE").items() if kern in k}, needle)
    if not fv or not wv:
        raise SystemExit(f"no FETCH_SIZE / WRITE_SIZE rows for {kern}")
    fetch_kb, write_kb = sum(fv) / len(fv), sum(wv) / len(wv)          # rocprofv3 reports both in KB
    note = ""
    if a.calib_dir:
        ck, cv = pick(counters(a.calib_dir, "FETCH_SIZE"), "epsm_tangent_kernel")
        if cv:
            known = 85 * a.paths                                    # o, d, dx, dy, p0, p1, p2 (7 x 12 B) + active (1 B) per path
            note = (f"; calibration in the same session: epsm_tangent_kernel reads {known / 1e9:.3f} GB, FETCH_SIZE reported "
                    f"{sum(cv) / len(cv) * 1024 / 1e9:.3f} GB")
    factor, fnote = 2.0, "x 2 (gfx950 half-count of wide coalesced reads)"
    if a.packed and a.gather_calib:
        gk, gv = pick(counters(a.gather_calib, "FETCH_SIZE"), "gather128")
        if gv:
            known = (1 << 23) * 128
            factor = known / (sum(gv) / len(gv) * 1024)
            fnote = (f"x {factor:.3f} (calibrated on tools/micro/gather128: per-lane 16-byte gathers of whole 128-byte records, "
                     f"{known / 1e9:.3f} GB read, FETCH_SIZE reported {sum(gv) / len(gv) * 1024 / 1e9:.3f} GB)")
    total = int(factor * fetch_kb * 1024 + write_kb * 1024)
    if a.rdreq_dir:
        n = {}
        for size in (32, 64, 128):
            _, v = pick({k: x for k, x in counters(a.rdreq_dir, f"TCC_EA0_RDREQ_{size}B_sum").items() if kern in k}, needle)
            n[size] = sum(v) / len(v) if v else None
        if all(x is not None for x in n.values()):
            read = 32 * n[32] + 64 * n[64] + 128 * n[128]
            total = int(read + write_kb * 1024)
            fnote = (f"tallied at 64 B per request; by request size (TCC_EA0_RDREQ_32B / _64B / _128B_sum = {n[32]:.4g} / {n[64]:.4g} / {n[128]:.4g}) "
                     f"the reads are {read / 1e9:.3f} GB -- used")
    sq = {}
    if a.sq_dir:
        for name, key in (("SQ_INSTS_VALU", "sq_insts_valu"), ("SQ_INSTS_SALU", "sq_insts_salu"), ("SQ_WAIT_ANY", "sq_wait_any"),
                          ("SQ_WAVE_CYCLES", "sq_wave_cycles")):
            _, v = pick({k: x for k, x in counters(a.sq_dir, name).items() if kern in k}, needle)
            if v:
                sq[key] = sum(v) / len(v)
    entry = {"kernel": "epsm_backward_pass_packed" if a.packed else "epsm_backward_pass", "paths": a.paths, "K": a.K, "variant": a.variant, "profile": a.profile,
             "hbm_bytes_per_launch": total, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
             "dispatches": [len(fv), len(wv)], "src_hash": bench.kernel_source_hash(),
             **sq,
             "source": f"profiles/{a.tag}_pmc_traffic.txt ({fk[:60]}...): FETCH_SIZE {fetch_kb:,.0f} KB {fnote} + "
                       f"WRITE_SIZE {write_kb:,.0f} KB (stores + float atomics, exact); separate rocprofv3 --pmc passes{note}"}
    path = os.path.join(ROOT, "profiles", "traffic.json")
    old = json.load(open(path)) if os.path.isfile(path) else []
    old = [e for e in old if not all(e.get(k) == entry[k] for k in ("kernel", "paths", "K", "variant", "profile"))]
    json.dump([entry] + old, open(path, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
