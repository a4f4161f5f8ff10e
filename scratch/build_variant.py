#!/usr/bin/env python3
"""Experiment builds: re-compiles the listed translation units with extra -D flags and links them with the regular objects into
scratch/variants/libwf_<name>.so (load it with WF_LIB=...).   usage: build_variant.py name "-DWF_NO_FENCE -DWF_XHALF=1" [unit ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from waveflow_amd import build as B  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2].split()
    units = sys.argv[3:] or ["wf_mfma_inst_d2.hip", "wf_mfma_inst_d2t2.hip"]
    B.build()
    out_dir = os.path.join(ROOT, "scratch", "variants")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for s in B.SOURCES:
        o = os.path.join(B.OBJ, s + ".o")
        if s in units:
            o = os.path.join(out_dir, f"{name}_{s}.o")
            cmd = [B._hipcc()] + B.FLAGS + (B.MFMA_FLAGS if ("mfma" in s or "etile" in s) else []) + flags + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(B.CSRC, s), "-o", o]
            if "--save-temps" in os.environ.get("WF_VARIANT_OPTS", ""):
                cmd += ["-save-temps=obj"]
            subprocess.run(cmd, check=True, cwd=out_dir)
        objs.append(o)
    lib = os.path.join(out_dir, f"libwf_{name}.so")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()
