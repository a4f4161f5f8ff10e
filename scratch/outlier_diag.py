"""After 12 000 large-batch training steps through the matrix-core paths: local energies of staged-sampler walkers by the tile kernel and by the wave kernel --
where do they differ, and are there walkers one sampler draws that the other does not?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from waveflow_amd import vqmc
from waveflow_amd.utils import physics
t = vqmc.ModelTrainer(system_name="He", learning_rate=1e-3, box_length=10, num_epochs=int(os.environ.get("EPOCHS", 12000)), batch_size=32768, log_every=10 ** 9)
t.save_dir = "/tmp/wf_outlier"
t.exact_sampler = True
params, loss = t.start_training(verbose=False)
m = t.psi.model
m.ensure_params(params)
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
n = 1 << 18
def eloc(x, tile):
    os.environ["WF_ENERGY_TILE_MIN"] = "1" if tile else "0"
    h, p = m.hamiltonian(x, protons, return_psi=True)
    del os.environ["WF_ENERGY_TILE_MIN"]
    return (h / (p + 1e-8)).double(), p.double()
for name, env in (("staged", "16384"), ("wave", "0")):
    os.environ["WF_SAMPLE_TILE_MIN"] = env
    os.environ["WF_WAVE_SAMPLE_MAX"] = "100000000"
    x, lat = m.sample(77, n, return_latent=True, exact=True)
    et, pt = eloc(x, True)
    ew, pw = eloc(x, False)
    d = (et - ew).abs()
    print(f"{name} sampler: <E_L> tile {et.mean().item():+.5f} wave {ew.mean().item():+.5f}; max |E_L| tile {et.abs().max().item():.3e} wave {ew.abs().max().item():.3e}; "
          f"max |diff| {d.max().item():.3e}; walkers with |diff| > 1: {(d > 1).sum().item()}; min |psi| {pt.abs().min().item():.3e}; x range [{x.min().item():.4f}, {x.max().item():.4f}]; "
          f"latent range [{lat.min().item():.6f}, {lat.max().item():.6f}]; unsorted {(x[:, 0] > x[:, 1]).sum().item()}")
    idx = torch.topk(et.abs(), 5).indices
    for i in idx.tolist():
        print(f"     walker {i}: x {x[i].tolist()} latent {lat[i].tolist()} E_L tile {et[i].item():+.4e} wave {ew[i].item():+.4e} psi {pt[i].item():.3e}")
