#!/usr/bin/env python3
"""Round 4, VERDICT item 1a on the GPU: where does k_mfma's deviation from the fp32 reference come from?
Experiment builds of the kernel (scratch/variants/libwf_<name>.so: one class of hardware approximations at a time replaced by fp64 arithmetic, the
centred first hidden layer on / off, the fourth product) evaluate C3's 2^20 walkers (and C2's 256); the parity statistics of oracle/parity_stats.py
are printed per build.   usage: r04_parity_variants.py name1 name2 ...   ('default' = the shipped library)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import bench
m, flat = bench.he_model(os.environ.get("WF_KERNEL", "mfma"))
x = torch.from_numpy(np.load(sys.argv[1])).cuda()
np.save(sys.argv[2], m.log_pdf(x).cpu().numpy())
''' % ROOT


def main():
    import oracle
    from oracle import parity_stats
    out = os.path.join(ROOT, "gpurun_out", "r04_parity_variants")
    os.makedirs(out, exist_ok=True)
    flat = np.load(os.path.join(ROOT, "tests", "golden", "he_checkpoint.npz"))["flat"]
    om = oracle.he_model(10.0)
    thr = max(1, min(16, len(os.sched_getaffinity(0))))
    xb = np.sort(np.random.default_rng(1234).uniform(-10, 10, size=(1 << 20, 2)).astype(np.float32), -1)
    np.save(os.path.join(out, "x.npy"), xb)
    lp32, _, _ = om.log_pdf_cond(flat, xb, threads=thr)
    lp64, cond, _ = om.log_pdf_cond(flat, xb, threads=thr, f64=True)
    for name in sys.argv[1:]:
        env = dict(os.environ)
        kern = "mfma"
        if name in ("scalar", "wave"):
            kern = name
        elif name != "default":
            env["WF_LIB"] = os.path.join(ROOT, "scratch", "variants", f"libwf_{name}.so")
            env["WF_LIB_EXPERIMENT"] = "1"
        env["WF_KERNEL"] = kern
        r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(out, "x.npy"), os.path.join(out, "lp.npy")], env=env, capture_output=True, text=True)
        if r.returncode:
            print(name, "FAILED", r.stderr[-500:])
            continue
        st = parity_stats.stats(np.load(os.path.join(out, "lp.npy")), lp32, lp64, cond)
        print(parity_stats.line(f"C3 2^20 {name}", st), flush=True)


if __name__ == "__main__":
    main()
