"""rocprofv3 target: 6 calls of wf_vqmc_loss_grad on 2^17 sampled He walkers through the matrix-core gradient path."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd.utils import physics
m, flat = bench.he_model("auto")
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
x = m.sample(11, int(os.environ.get("B", 1 << 17)), exact=True)
for _ in range(6):
    m.vqmc_loss_grad(x, protons, -1.8)
torch.cuda.synchronize()
