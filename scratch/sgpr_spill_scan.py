"""Which kernels of the built library spill SGPRs into VGPR lanes (v_writelane_b32 / v_readlane_b32 pairs)?  usage: sgpr_spill_scan.py [lib]"""
import os, re, subprocess, sys, tempfile, shutil
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "waveflow_amd", "libwaveflow_hip.so")
od = "/opt/rocm/lib/llvm/bin/llvm-objdump"
sym = re.compile(r"^[0-9a-f]+ <(.+)>:\s*$")
out = {}
with tempfile.TemporaryDirectory() as tmp:
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([od, "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        text = subprocess.run([od, "-d", os.path.join(tmp, f)], capture_output=True, text=True).stdout
        cur = None
        for line in text.splitlines():
            m = sym.match(line)
            if m:
                cur = m.group(1); out.setdefault(cur, [0, 0, 0]); continue
            if cur is None: continue
            if "v_writelane_b32" in line: out[cur][0] += 1
            elif "v_readlane_b32" in line: out[cur][1] += 1
            elif "s_cbranch_exec" in line: out[cur][2] += 1
dem = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True).stdout.splitlines()
rows = sorted(((v[0], v[1], v[2], d) for (k, v), d in zip(out.items(), dem) if v[0] > 0), reverse=True)
print("writelane readlane exec-branches kernel")
for w, r, e, d in rows:
    print(f"{w:9d} {r:8d} {e:13d} {d[:110]}")
print(len(rows), "of", len(out), "functions spill SGPRs to VGPR lanes")
