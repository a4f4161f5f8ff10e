"""H psi of 2^20 He walkers through the tile path with the library named by WF_LIB (or the default one)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import sorted_walkers
import bench
from waveflow_amd.utils import physics
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
xb = torch.as_tensor(sorted_walkers(1 << 20, 2, 10.0, 1234)).cuda()
os.environ["WF_ENERGY_TILE_MIN"] = "1"
m, flat = bench.he_model("auto")
for _ in range(30): m.hamiltonian(xb, protons)
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); h = m.hamiltonian(xb, protons); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(os.environ.get("WF_LIB", "default"), "H psi 2^20: median %.4f ms  min %.4f ms  -> %.3e walkers/s  checksum %.9e" % (np.median(ts), np.min(ts), (1 << 20) / np.median(ts) * 1e3, float(h.double().sum())))
