"""rocprofv3 target: 6 calls of wf_vqmc_loss_grad on 2^17 walkers of the 33-knot model (KN=23: the shipped He checkpoint) through the matrix-core gradient path."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd.utils import physics
kn = int(os.environ.get("KN", 33))
m = bench.seeded_model(2, 33, "auto") if kn == 33 else bench.he_model("auto")[0]
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
x = bench.walkers(1 << 17, 1234).cuda()
for _ in range(6):
    m.vqmc_loss_grad(x, protons, -1.8)
torch.cuda.synchronize()
