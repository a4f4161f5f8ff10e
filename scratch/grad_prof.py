"""rocprofv3 target: 10 calls of wf_vqmc_loss_grad on 2^17 He walkers (the loss_grad_2pow17 leg of bench.py)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd.utils import physics
m, flat = bench.he_model("auto")
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
x = bench.walkers(1 << 17, 1234).cuda()
for _ in range(10):
    m.vqmc_loss_grad(x, protons, -1.8)
torch.cuda.synchronize()
