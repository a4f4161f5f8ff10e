"""A/B of builds on the three k_mfma shapes of the bench (He 23 knots 2^20, 33 knots 2^20, 8-electron chain 2^18): interleaved rounds.
usage: r04_time3.py lib1 lib2 ... (names under scratch/variants, or 'default')"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import bench
def t(m, x):
    for _ in range(10): m.log_pdf(x)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = m.log_pdf(x); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return np.median(ts), int(np.frombuffer(y.cpu().numpy().tobytes(), dtype=np.uint32).astype(np.uint64).sum())
m, _ = bench.he_model("mfma"); print("RES he23", *t(m, bench.walkers(1 << 20, 1234).cuda()))
m = bench.seeded_model(2, 33, "mfma"); print("RES he33", *t(m, bench.sorted_uniform(1 << 20, 2, 1234).cuda()))
m = bench.seeded_model(8, 23, "mfma"); print("RES c4", *t(m, bench.sorted_uniform(1 << 18, 8, 1234).cuda()))
''' % ROOT
res = {}
for r in range(int(os.environ.get("ROUNDS", "3"))):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default":
            env["WF_LIB"] = os.path.join(ROOT, "scratch", "variants", f"libwf_{lib}.so"); env["WF_LIB_EXPERIMENT"] = "1"
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if line.startswith("RES"):
                p = line.split(); res.setdefault((lib, p[1]), []).append((float(p[2]), p[3]))
        if out.returncode: print(lib, "FAILED", out.stderr[-300:])
for (lib, cfg), v in res.items():
    print(f"{lib:14s} {cfg:5s}: median of round medians {np.median([a for a, _ in v]):.4f} ms  rounds {['%.4f' % a for a, _ in v]}  checksum {v[0][1]}")
