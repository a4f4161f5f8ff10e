#!/usr/bin/env python3
"""Spill traffic and instruction mix of k_ebwd per phase, from the assembly of a -DWF_MARKS build (scratch/build_variant.py marks "-DWF_MARKS" wf_kernels_etile.hip
with WF_VARIANT_OPTS=--save-temps).  usage: r04_spill_phases.py <file.s> [kernel-name-regex]"""
import re, sys, collections
path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_ebwd")
cur_kernel, phase = None, None
stats = collections.OrderedDict()
for line in open(path):
    t = line.strip()
    m = re.match(r"^(_Z\w+):", t)
    if m:
        cur_kernel = m.group(1) if re.search(pat, m.group(1)) else None
        phase = "prologue"
        continue
    if cur_kernel is None:
        continue
    if t.startswith(".Lfunc_end"):
        cur_kernel = None
        continue
    m = re.match(r"; WF_MARK (\w+)", t)
    if m:
        phase = "after " + m.group(1)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    d = stats.setdefault((cur_kernel, phase), collections.Counter())
    d["all"] += 1
    if op.startswith("scratch_store"): d["spill_st"] += 1
    elif op.startswith("scratch_load"): d["spill_ld"] += 1
    elif op.startswith("v_mfma"): d["mfma"] += 1
    elif op.startswith("v_accvgpr"): d["acc_mov"] += 1
    elif op.startswith("v_"): d["valu"] += 1
    elif op.startswith("ds_"): d["lds"] += 1
    elif op.startswith(("global_", "buffer_")): d["vmem"] += 1
    elif op.startswith("s_waitcnt"): d["waitcnt"] += 1
    elif op.startswith("s_nop"): d["nop"] += 1
last = None
for (k, ph), d in stats.items():
    if k != last:
        print("\n" + k); last = k
        print(f"  {'phase':24s} {'all':>6s} {'valu':>6s} {'mfma':>5s} {'accmov':>6s} {'lds':>5s} {'vmem':>5s} {'sp_st':>5s} {'sp_ld':>5s} {'wait':>5s} {'nop':>5s}")
    print(f"  {ph:24s} {d['all']:6d} {d['valu']:6d} {d['mfma']:5d} {d['acc_mov']:6d} {d['lds']:5d} {d['vmem']:5d} {d['spill_st']:5d} {d['spill_ld']:5d} {d['waitcnt']:5d} {d['nop']:5d}")
