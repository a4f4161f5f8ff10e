#!/usr/bin/env python3
"""Register / scratch / LDS use of the kernels of a built library, from the code-object metadata (no GPU needed).
usage: kernel_resources.py [lib.so] [name-regex]      (default: waveflow_amd/libwaveflow_hip.so, 'k_mfma')"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LL = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(lib, pattern):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(lib, os.path.join(tmp, "l.so"))
        subprocess.run([f"{LL}/llvm-objdump", "--offloading", "l.so"], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([f"{LL}/llvm-readelf", "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name or not re.search(pattern, name.group(1)):
                    continue
                g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) if re.search(rf"\.{k}:\s+(\d+)", blk) else -1  # noqa: E731
                agpr = int(re.match(r"\s*(\d+)", blk).group(1))
                dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
                out.append((dem.split("(")[0], g("vgpr_count"), agpr, g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"),
                            g("vgpr_spill_count")))
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "waveflow_amd", "libwaveflow_hip.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else "k_mfma"
    print(f"{'kernel':70s} vgpr agpr sgpr scratch lds spills")
    for r in sorted(set(resources(lib, pat))):
        print(f"{r[0][-70:]:70s} {r[1]:4d} {r[2]:4d} {r[3]:4d} {r[4]:7d} {r[5]:5d} {r[6]:4d}")
