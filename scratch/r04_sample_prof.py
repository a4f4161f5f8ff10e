"""rocprofv3 target: 8 calls of the staged sampler, 2^17 exact draws of the shipped He model"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
m, _ = bench.he_model("auto")
for s in range(8):
    m.sample(100 + s, 1 << 17, exact=True)
torch.cuda.synchronize()
