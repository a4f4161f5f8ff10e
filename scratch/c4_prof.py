"""rocprofv3 target: log_pdf of the 8-electron chain (BASELINE config 4) on 2^18 walkers, 30 launches."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
m8 = bench.seeded_model(8, 23, "mfma")
x = bench.sorted_uniform(1 << 18, 8, 1234).cuda()
for _ in range(30):
    m8.log_pdf(x)
torch.cuda.synchronize()
