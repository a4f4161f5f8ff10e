"""C4's kernel k_mfma<8,1,W,1>: rolled dimension loops (default library), unrolled (round 3's form), and the 12- / 16-wave builds of the rolled form.
Interleaved rounds in one run; the outputs of every build must be the same bits.
Record of the run behind profiles/r04_c4_rolled_loops_and_waves.txt: the default library was then built rolled (now: -DWF_D8_ROLLED), the variants
were scratch/variants/libwf_d8unrolled.so (today's default) and libwf_d8waves.so (-DWF_D8_ROLLED -DWF_D8_WAVES_ALL)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import bench
m = bench.seeded_model(int(os.environ.get("DD", "8")), 23, "mfma")
x = bench.sorted_uniform(1 << 18, int(os.environ.get("DD", "8")), 1234).cuda()
for _ in range(10): m.log_pdf(x)
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = m.log_pdf(x); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("RES", np.median(ts), np.min(ts), float(y.double().sum()), int(np.frombuffer(y.cpu().numpy().tobytes(), dtype=np.uint32).astype(np.uint64).sum()))
''' % ROOT
cases = [("rolled 8 waves (default lib)", None, None), ("unrolled 8 waves (round 3)", "d8unrolled", None), ("rolled 12 waves", "d8waves", "12"), ("rolled 16 waves", "d8waves", "16")]
res = {}
for r in range(3):
    for name, lib, waves in cases:
        env = dict(os.environ)
        env.pop("WF_MFMA_WAVES", None)
        if lib:
            env["WF_LIB"] = os.path.join(ROOT, "scratch", "variants", f"libwf_{lib}.so"); env["WF_LIB_EXPERIMENT"] = "1"
        if waves:
            env["WF_MFMA_WAVES"] = waves
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if line.startswith("RES"):
                p = line.split(); res.setdefault(name, []).append((float(p[1]), float(p[2]), p[3], p[4]))
        if out.returncode: print(name, "FAILED", out.stderr[-300:])
for name, v in res.items():
    print(f"{name:32s}: median of round medians {np.median([a for a, *_ in v]):.4f} ms  min {min(b for _, b, *_ in v):.4f} ms   checksum {v[0][3]}  rounds {['%.4f' % a for a, *_ in v]}")
for dd in ("5", "6", "7"):
    for name, lib in (("rolled", None), ("unrolled", "d8unrolled")):
        env = dict(os.environ, DD=dd)
        if lib:
            env["WF_LIB"] = os.path.join(ROOT, "scratch", "variants", f"libwf_{lib}.so"); env["WF_LIB_EXPERIMENT"] = "1"
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("D =", dd, name, [l for l in out.stdout.splitlines() if l.startswith("RES")])
