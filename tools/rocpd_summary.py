#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd SQLite output) runs into small text tables for profiles/.

    python tools/rocpd_summary.py trace <trace_results.db>            # = --kernel-trace --stats
    python tools/rocpd_summary.py pmc   <pmc_results.db> [...]        # one DB per --pmc pass

PMC values are summed over the per-XCD/SE instances rocprofv3 reports and averaged per launch.
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 definition); on gfx950 FETCH_SIZE under-reports wide
coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) - the 'x2' column applies that correction.
"""
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def trace(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"{'kernel':44s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in rows:
        print(f"{short(name)[:44]:44s} {calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")


def pmc(dbs):
    per = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    dur = defaultdict(lambda: [0.0, 0])
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        q = "select kernel_name, counter_name, dispatch_id, value, duration from counters_collection"
        seen = set()
        for kname, cname, disp, val, d in cur.execute(q):
            k = short(kname)
            e = per[k][cname]
            e[0] += float(val)
            e[1].add((db, disp))
            if (db, disp) not in seen:
                seen.add((db, disp))
                dur[k][0] += d
                dur[k][1] += 1
    counters = sorted({c for k in per for c in per[k]})
    print("# rocprofv3 --pmc summary (per-launch averages) of " + ", ".join(dbs))
    hdr = f"{'kernel':30s} {'launches':>8s} {'avg_us':>9s} " + " ".join(f"{c[:24]:>24s}" for c in counters)
    if "FETCH_SIZE" in counters:
        hdr += f" {'FETCH_x2_MB':>12s}"
    util = "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "GRBM_GUI_ACTIVE" in counters
    if util:
        hdr += f" {'mfma_busy_frac':>14s}"     # busy cycles per SIMD (1024 SIMDs) / active cycles per XCD (8)
    print(hdr)
    for k in sorted(per, key=lambda k: -dur[k][0]):
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        n = max(len(per[k][c][1]) for c in per[k])
        line = f"{k[:30]:30s} {n:8d} {dur[k][0] / max(1, dur[k][1]) / 1e3:9.1f} "
        line += " ".join(f"{per[k][c][0] / max(1, len(per[k][c][1])):24.1f}" if c in per[k] else f"{'-':>24s}"
                         for c in counters)
        if "FETCH_SIZE" in per[k]:
            f = per[k]["FETCH_SIZE"]
            line += f" {2 * f[0] / max(1, len(f[1])) / 1024:12.2f}"
        if util and "GRBM_GUI_ACTIVE" in per[k]:
            mb, ga = per[k]["SQ_VALU_MFMA_BUSY_CYCLES"], per[k]["GRBM_GUI_ACTIVE"]
            line += f" {(mb[0] / max(1, len(mb[1])) / 1024) / (ga[0] / max(1, len(ga[1])) / 8):14.3f}"
        print(line)


if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    if sys.argv[1] == "trace":
        trace(sys.argv[2])
    else:
        pmc(sys.argv[2:])
