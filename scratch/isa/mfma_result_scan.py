#!/usr/bin/env python3
"""For every MFMA whose result is next read by a non-MFMA instruction: the instructions in between (what the compiler counted as wait
states).  usage: mfma_result_scan.py file.s [kernel-substring]"""
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_stats import kernels, classify  # noqa: E402
from war_scan import regs  # noqa: E402


def all_regs(u):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|v(\d+)", u):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, lines in kernels(path).items():
        if sub not in name:
            continue
        ins = []
        for ln in lines:
            t = ln.split(";")[0].strip()
            if not t or t.startswith(".") or t.endswith(":"):
                continue
            ins.append(t)
        print(name)
        for i, t in enumerate(ins):
            if not t.startswith("v_mfma"):
                continue
            d = regs(t.split(None, 1)[1].split(",")[0])
            between = []
            for j in range(i + 1, min(i + 40, len(ins))):
                u = ins[j]
                if u.startswith("v_mfma") and regs(u.split(None, 1)[1].split(",")[3].strip()) == d:
                    break   # accumulate chain continues
                if all_regs(u) & d and not u.startswith("s_"):
                    kinds = {}
                    states = 0
                    for b in between:
                        op = b.split()[0]
                        m = re.match(r"s_nop\s+(\d+)", b)
                        n = int(m.group(1)) + 1 if m else 1
                        states += n
                        c = "nop" if m else classify(op)
                        kinds[c] = kinds.get(c, 0) + n
                    print(f"  [{i}] {t.split()[0]} -> [{j}] {u[:60]:60s} states {states:3d} {kinds}")
                    break
                between.append(u)


if __name__ == "__main__":
    main()
