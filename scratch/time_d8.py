import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
m8 = bench.seeded_model(8, 23, "mfma")
x8 = bench.sorted_uniform(1 << 18, 8, 1234).cuda()
for w in ("8", "16"):
    os.environ["WF_MFMA_WAVES"] = w
    print("D=8 2^18 waves", w, bench.kernel_ms(m8, x8, n=20, warm=5), "ms")
