#!/usr/bin/env python3
"""Instruction mix of kernels in a built library (whole kernel, static).  usage: isa_count.py lib.so name-regex"""
import os, re, shutil, subprocess, sys, tempfile, collections
LL = "/opt/rocm/lib/llvm/bin"
lib, pat = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as tmp:
    shutil.copy(lib, os.path.join(tmp, "l.so"))
    subprocess.run([f"{LL}/llvm-objdump", "--offloading", "l.so"], cwd=tmp, check=True, capture_output=True)
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f: continue
        dis = subprocess.run([f"{LL}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(tmp, f)], capture_output=True, text=True).stdout
        cur = None; stats = collections.OrderedDict()
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1) if re.search(pat, m.group(1)) and not m.group(1).startswith("__") else None
                continue
            if cur is None: continue
            t = line.strip().split()
            if not t: continue
            op = t[0]
            d = stats.setdefault(cur, collections.Counter())
            d["all"] += 1
            for k, pre in (("sp_st", "scratch_store"), ("sp_ld", "scratch_load"), ("mfma", "v_mfma"), ("accmov", "v_accvgpr"), ("lds", "ds_"), ("vmem", "global_"), ("wait", "s_waitcnt"), ("nop", "s_nop")):
                if op.startswith(pre): d[k] += 1; break
            else:
                if op.startswith("v_"): d["valu"] += 1
                if op in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32"): d["trans"] += 1
        for k, d in stats.items():
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:70]
            print(f"{name:70s} all {d['all']:6d} valu {d['valu']:6d} trans {d['trans']:5d} mfma {d['mfma']:4d} accmov {d['accmov']:5d} lds {d['lds']:4d} vmem {d['vmem']:4d} sp_st {d['sp_st']:4d} sp_ld {d['sp_ld']:4d} wait {d['wait']:4d} nop {d['nop']:4d}")
