#!/usr/bin/env python3
"""Per-phase vector-instruction budget of the headline kernel k_mfma<2,1,16,1,false,true> from its disassembly.
usage: r04_valu_budget.py file.s   (hipcc -S --cuda-device-only of wf_mfma_inst_d2.hip)
The tile loop of the SPEC instantiation is a handful of basic blocks: per flow net [hidden layers | output block + sigmoids + record requests |
spline dots (one of two variants: with / without wrapped indices) | finish], then the prior head [hidden | output + scale / split | ob_to_b + norm |
rows dot | finish].  Every instruction is priced with the issue rates measured at four waves per SIMD (v_exp / v_rcp / v_log / v_rsq and
v_fma_mix* 6.4 cycles, v_cvt* / v_floor / v_ceil / v_med3 / v_ldexp / v_frexp 3.4, other VALU 2.0, an MFMA holds the issue port 8)."""
import re
import sys
from collections import OrderedDict

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_")
SLOW = ("v_cvt_", "v_floor_", "v_ceil_", "v_med3_", "v_ldexp_", "v_frexp_", "v_permlane")


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_fma_mix"):
        return "mix"
    if op.startswith(SLOW):
        return "slow"
    if op.startswith("v_"):
        return "plain"
    return None


COST = {"mfma": 8.0, "trans": 6.4, "mix": 6.4, "slow": 3.4, "plain": 2.0}


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2wf4mfma6k_mfmaILi2ELi1ELi16ELi1ELb0ELb1E") and l.rstrip().endswith("Pi"))
    blocks, cur = OrderedDict(), "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end") or l.startswith("\t.end_amdhsa_kernel"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        t = l.strip().split()
        if t and not t[0].startswith((";", ".")):
            blocks[cur].append(t[0])
    rows = []
    for name, ops in blocks.items():
        c = {k: 0 for k in COST}
        for op in ops:
            k = cls(op)
            if k:
                c[k] += 1
        n = sum(c.values())
        if n >= 20:
            rows.append((name, c, n, sum(COST[k] * v for k, v in c.items())))
    print("%-12s %6s %6s %6s %6s %6s %7s %9s" % ("block", "mfma", "trans", "mix", "slow", "plain", "instr", "cycles"))
    for name, c, n, cyc in rows:
        print("%-12s %6d %6d %6d %6d %6d %7d %9.0f" % (name, c["mfma"], c["trans"], c["mix"], c["slow"], c["plain"], n, cyc))


if __name__ == "__main__":
    main()
