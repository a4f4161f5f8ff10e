#!/usr/bin/env python3
"""Scans a gfx950 .s for writes to a VGPR that a preceding MFMA reads as A / B / C within `window` instructions (WAR candidates).
usage: war_scan.py file.s [kernel-substring] [window]"""
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_stats import kernels, classify  # noqa: E402


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    return set()


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    window = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    for name, lines in kernels(path).items():
        if sub not in name:
            continue
        ins = []
        for ln in lines:
            t = ln.split(";")[0].strip()
            if not t or t.startswith(".") or t.endswith(":"):
                continue
            ins.append(t)
        print(name, len(ins), "instructions")
        hits = {}
        for i, t in enumerate(ins):
            if not t.startswith("v_mfma"):
                continue
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            srcs = {"A": regs(ops[1]), "B": regs(ops[2]), "C": regs(ops[3]) - regs(ops[0])}
            states = 0
            for j in range(i + 1, min(i + 1 + window, len(ins))):
                u = ins[j]
                op = u.split()[0]
                c = classify(op)
                if c in ("valu", "trans", "xlane", "lds", "vmem") and not op.startswith(("ds_write", "global_store", "buffer_store")):
                    dst = regs(u.split(None, 1)[1].split(",")[0]) if len(u.split(None, 1)) > 1 else set()
                    if op.startswith("v_permlane32_swap") or op.startswith("v_swap"):
                        dst |= regs(u.split(None, 1)[1].split(",")[1])
                    for k, s in srcs.items():
                        if dst & s:
                            key = (k, states, c)
                            hits.setdefault(key, []).append((i, t, j, u))
                if op.startswith("v_mfma"):
                    pass
                m = re.match(r"s_nop\s+(\d+)", u)
                states += (int(m.group(1)) + 1) if m else 1
        for key in sorted(hits):
            print(f"  src{key[0]} overwritten after {key[1]} wait states by {key[2]}: {len(hits[key])} sites")
            for (i, t, j, u) in hits[key][:3]:
                print(f"      [{i}] {t}\n      [{j}] {u}")


if __name__ == "__main__":
    main()
