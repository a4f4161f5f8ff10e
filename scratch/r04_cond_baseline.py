"""rocprofv3 target: the launch-per-net form of H psi (WF_ENERGY_FUSED=0) -- times the library's 32-walker, three-channel conditioner launches
(k_etile_cond<false, 1, 3>) for scratch/ubench3/cond16.hip"""
import os, sys, torch
os.environ["WF_ENERGY_FUSED"] = "0"
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd.utils import physics
m, _ = bench.he_model("auto")
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
x = bench.walkers(int(os.environ.get("B", 1 << 17)), 4321).cuda()
for _ in range(12):
    m.hamiltonian(x, protons)
torch.cuda.synchronize()
