#!/usr/bin/env python3
"""Static instruction statistics of the kernels in a hipcc -save-temps .s file (per basic block and per kernel).
usage: isa_stats.py file.s [kernel-substring] [--blocks]"""
import re
import sys
from collections import Counter, OrderedDict

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_permlane") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "xlane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(path):
    cur, out = None, OrderedDict()
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        out[cur].append(line.rstrip("\n"))
    return out


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    blocks = "--blocks" in sys.argv
    for name, lines in kernels(path).items():
        if sub not in name:
            continue
        tot = Counter()
        blk, bname = Counter(), "entry"
        per_block = []
        for ln in lines:
            m = re.match(r"^(\.LBB\w+):", ln)
            if m:
                per_block.append((bname, blk))
                blk, bname = Counter(), m.group(1)
                continue
            t = ln.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.startswith("//"):
                continue
            op = t.split()[0]
            c = classify(op)
            tot[c] += 1
            blk[c] += 1
            if c == "nop":
                mm = re.match(r"s_nop\s+(\d+)", t)
                tot["nop_states"] += int(mm.group(1)) + 1 if mm else 1
                blk["nop_states"] += int(mm.group(1)) + 1 if mm else 1
        per_block.append((bname, blk))
        print(name)
        print("   total:", dict(sorted(tot.items())))
        if blocks:
            for b, c in per_block:
                n = sum(v for k, v in c.items() if k != "nop_states")
                if n >= 20:
                    print(f"   {b:14s} n={n:5d}", dict(sorted(c.items())))


if __name__ == "__main__":
    main()
