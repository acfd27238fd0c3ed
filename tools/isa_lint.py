"""ISA lint for one hazard the compiler does not cover on gfx950 (found in round 4): the data registers of
an LDS store with MORE than 64 bits of data in flight (ds_write2_b64 / ds_write2st64_b64 / ds_write_b96 / ds_write_b128) are
read by the LDS path for several cycles after the instruction issues; a VALU instruction that overwrites one of them within
the next few issue slots corrupts the stored value (measured: ds_write2_b64 v61, v[52:53], v[40:41] followed by
v_add_u32 v40, ... stored the ADDRESS in the hi plane of the window).  For 16-byte VMEM stores the compiler inserts an s_nop;
for LDS stores it does not.

usage: python tools/isa_lint.py            (compiles every library source to gfx950 assembly and scans it)
Reports every (kernel, line) where a wide LDS store is followed, within WINDOW instructions and before a wait-state-
consuming s_nop / s_waitcnt, by a VALU instruction writing one of its data registers BEYOND the first two dwords."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WIDE = ("ds_write2_b64", "ds_write2st64_b64", "ds_write_b96", "ds_write_b128")
WINDOW = 2            # issue slots after the store in which a write to its data registers is flagged


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(text, src):
    out, kernel = [], "?"
    lines = text.split("\n")
    code = []
    for ln in lines:
        s = ln.strip()
        if s.endswith(":") and not s.startswith("."):
            kernel = s[:-1]
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        code.append((kernel, s.split(";")[0].strip()))
    for i, (k, ins) in enumerate(code):
        op, _, args = ins.partition(" ")
        if op not in WIDE:
            continue
        toks = [t.strip() for t in args.split(",")]
        order = []
        for t in toks[1:]:                       # toks[0] is the address
            order += sorted(regs(t.split()[0]))
        # The LDS path takes the address and then the data dwords in order, two cycles each: the first two data dwords are
        # gone before the next instruction can write them (every shipped kernel has such pairs and is bit-exact against
        # its oracle); the LATER dwords are the exposed ones.
        data = set(order[2:])
        slots = 0
        for k2, nxt in code[i + 1:i + 1 + 6]:
            o2, _, a2 = nxt.partition(" ")
            if o2.startswith(("s_nop", "s_waitcnt", "s_barrier")):
                break                            # wait states / a drain: the LDS has taken its operands
            if o2.startswith("s_"):
                slots += 1
                continue
            # (an MFMA writes its destination at the END of its passes - tens of cycles later - and a load when it returns)
            dst = regs(a2.split(",")[0].strip()) if o2.startswith("v_") and not o2.startswith(("v_cmp", "v_mfma")) else set()
            if dst & data:
                out.append((src, k, ins, nxt))
                break
            slots += 1
            if slots >= WINDOW:
                break
    return out


def main():
    from cmgan_amd import build as B

    def job(j):
        src, extra = j
        with tempfile.TemporaryDirectory() as d:
            o = os.path.join(d, "k.s")
            subprocess.run([B.HIPCC, *[f for f in B.FLAGS if f != "-fPIC"], *extra, "-S", "--cuda-device-only",
                            os.path.join(B.CSRC, src), "-o", o], check=True, stderr=subprocess.DEVNULL)
            return scan(open(o).read(), src + (" [x1]" if extra else ""))
    jobs = [(s, []) for s in B.SOURCES] + [(s, B.X1_FLAGS) for s in B.X1_SOURCES]
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        hits = [h for r in ex.map(job, jobs) for h in r]
    for src, k, a, b in hits:
        print(f"{src}: {k[:60]}\n    {a}\n    {b}")
    print(f"{len(hits)} hazard candidate(s)")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
