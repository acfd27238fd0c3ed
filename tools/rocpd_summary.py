#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd SQLite output) runs into small text tables for profiles/.

    python tools/rocpd_summary.py trace <trace_results.db>            # = --kernel-trace --stats
    python tools/rocpd_summary.py pmc   <pmc_results.db> [...]        # one DB per --pmc pass
    python tools/rocpd_summary.py traffic <FETCH_SIZE db> <WRITE_SIZE db> [bench.json]
                                          # JSON: HBM bytes per launch per kernel label, stamped with the csrc digest /
                                          # workload / batch of the bench line produced in the SAME GPU session

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


# mangled-name fragment -> the label bench.py's profiler uses for that kernel
LABELS = [("conv3x_kernelILi2ELi64", "conv_dense"), ("conv3x_kernelILi1ELi128", "conv_subpixel"),
          ("conv3x_kernelILi1ELi64", "conv_1x3"), ("conv3_kernelILi2ELi64", "conv_dense"),
          ("conv3_kernelILi1ELi128", "conv_subpixel"), ("conv3_kernelILi1ELi64", "conv_1x3"),
          ("attn_sp_out_x3_kernel", "attn_out"), ("attn32_out_x3_kernel", "attn_out"), ("qkv32_x3_kernel", "qkv"),
          ("attn_out_x3_kernel", "attn_out"), ("attn_x3_kernel", "attn"), ("attn_kernel", "attn"), ("dwpw2s_x3_kernel", "dwpw2"), ("dwpw2t_x3_kernel", "dwpw2"), ("dwpw2_x3_kernel", "dwpw2"),
          ("ffn32_x3_kernelILb1", "ffn_post"), ("ffn32_x3_kernelILb0", "ffn"),
          ("ffn_x3_kernelILb1", "ffn_post"), ("ffn_x3_kernelILb0", "ffn"), ("ffn_kernelILb1", "ffn_post"),
          ("ffn_kernelILb0", "ffn"), ("qkv_x3_kernel", "qkv"), ("qkv_kernel", "qkv"),
          ("pw1glu_x3_kernel", "pw1glu"), ("pw1glu_kernel", "pw1glu"), ("outproj_x3_kernel", "outproj"),
          ("outproj_kernel", "outproj"), ("dwconv_kernel", "dwconv"), ("pw2_kernel", "pw2"),
          ("stft_compress_kernel", "stft_compress"), ("uncompress_irfft_kernel", "uncompress_irfft"),
          ("tail_proj_kernel", "tail_proj")]


def traffic(fetch_db, write_db, bench_json=None):
    """HBM bytes per launch (FETCH_SIZE with the gfx950 2x correction for 16 B/lane reads + WRITE_SIZE)."""
    import json
    out = {}
    for db, cname, scale in ((fetch_db, "FETCH_SIZE", 2.0), (write_db, "WRITE_SIZE", 1.0)):
        cur = sqlite3.connect(db).cursor()
        acc = defaultdict(lambda: [0.0, set()])
        q = "select kernel_name, dispatch_id, value from counters_collection where counter_name = ?"
        for kname, disp, val in cur.execute(q, (cname,)):
            label = next((lab for frag, lab in LABELS if frag in kname), None)
            if label:
                acc[label][0] += float(val)
                acc[label][1].add(disp)
        for label, (tot, disps) in acc.items():
            e = out.setdefault(label, {"launches": len(disps)})
            e["fetch_bytes" if cname == "FETCH_SIZE" else "write_bytes"] = round(scale * tot * 1024 / len(disps))
    for e in out.values():
        e["hbm_bytes"] = e.get("fetch_bytes", 0) + e.get("write_bytes", 0)
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, "
                     "FETCH x2 per MI355X_MICROARCH.md (gfx950 under-reports 16 B/lane reads)"}
    if bench_json:
        # bench.py only publishes these counters while the built kernel sources still hash to this digest
        with open(bench_json) as f:
            b = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])
        doc["csrc_digest"] = b["csrc_digest"]
        doc["workload"] = "48k" if "48 kHz" in b["config"]["workload"] else "16k"
        doc["batch"] = b["config"]["batch_per_gpu"]
    doc["per_launch"] = out
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    if sys.argv[1] == "trace":
        trace(sys.argv[2])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        pmc(sys.argv[2:])
