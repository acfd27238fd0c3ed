#!/usr/bin/env python3
"""Static instruction mix per basic block of one kernel of a gfx950 assembly dump (hipcc -S --cuda-device-only):
    tools/isa_blocks.py file.s <kernel-name-substring>
Used to find instruction overhead OUTSIDE the hot bodies (e.g. the attention's per-tile blocks: SGPR spills through
v_writelane / v_readlane, integer divisions, loop-invariant masks)."""
import re
import sys
from collections import Counter


def blocks(text, sub):
    m = re.search(r"^(\S*" + re.escape(sub) + r"\S*):.*?\n(.*?)\n\s+s_endpgm", text, re.S | re.M)
    if not m:
        raise SystemExit(f"kernel *{sub}* not found")
    out, cur = [], None
    for l in m.group(2).split("\n"):
        s = l.strip()
        if not s or s.startswith(";"):
            continue
        if re.match(r"^\.?[A-Za-z_0-9$.]+:", s):
            cur = [s.split(":")[0], Counter()]
            out.append(cur)
            continue
        if s.startswith("."):
            continue
        if cur is None:
            cur = ["entry", Counter()]
            out.append(cur)
        i = s.split()[0]
        kind = ("mfma" if i.startswith("v_mfma") else "valu" if i.startswith("v_") else "salu" if i.startswith("s_")
                else "lds" if i.startswith("ds_") else "vmem" if i.startswith(("buffer_", "global_")) else "other")
        cur[1][kind] += 1
        for tag, pat in (("div", "rcp_iflag"), ("exp", "v_exp"), ("wl", "writelane"), ("rl", "readlane"), ("mov", "v_mov")):
            if pat in i:
                cur[1][tag] += 1
    return m.group(1), out


if __name__ == "__main__":
    name, bl = blocks(open(sys.argv[1]).read(), sys.argv[2])
    print(name)
    tot = Counter()
    for b, c in bl:
        n = sum(c[k] for k in ("mfma", "valu", "salu", "lds", "vmem"))
        tot.update(c)
        if n >= 12:
            print(f"{b[:12]:>12} {n:5d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
    print("total", dict(tot))
