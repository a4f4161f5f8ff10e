#!/usr/bin/env python3
"""VALU / LDS / VMEM write -> MFMA read (A, B, C) distances in wait states.  usage: raw_scan.py file.s [kernel-substring] [window]"""
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_stats import kernels, classify  # noqa: E402
from war_scan import regs  # noqa: E402


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    window = int(sys.argv[3]) if len(sys.argv) > 3 else 4
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
        hits = {}
        for i, t in enumerate(ins):
            if not t.startswith("v_mfma"):
                continue
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            srcs = {"A": regs(ops[1]), "B": regs(ops[2]), "C": regs(ops[3])}
            states = 0
            for j in range(i - 1, max(i - 1 - window, -1), -1):
                u = ins[j]
                op = u.split()[0]
                c = classify(op)
                m = re.match(r"s_nop\s+(\d+)", u)
                if m:
                    states += int(m.group(1)) + 1
                    continue
                if c in ("valu", "trans", "xlane", "mfma") and len(u.split(None, 1)) > 1:
                    dst = regs(u.split(None, 1)[1].split(",")[0])
                    for k, s in srcs.items():
                        if dst & s and not (c == "mfma" and k == "C"):
                            kind = op if ("mix" in op or "sdwa" in u or "cvt_pk" in op or c == "mfma") else c
                            hits.setdefault((k, states, kind), []).append((j, u, i, t))
                states += 1
        for key in sorted(hits):
            print(f"  src{key[0]} written {key[1]} wait states before the MFMA by {key[2]}: {len(hits[key])} sites")
            for (j, u, i, t) in hits[key][:2]:
                print(f"      [{j}] {u}\n      [{i}] {t}")


if __name__ == "__main__":
    main()
