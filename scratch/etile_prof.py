import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import sorted_walkers
import bench
from waveflow_amd.utils import physics
m, flat = bench.he_model("auto")
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
xb = torch.as_tensor(sorted_walkers(1 << 20, 2, 10.0, 1234)).cuda()
os.environ["WF_ENERGY_TILE_MIN"] = "1"
for _ in range(12): m.hamiltonian(xb, protons)
torch.cuda.synchronize()
