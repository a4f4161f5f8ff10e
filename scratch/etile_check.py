"""H psi through the tile path (wf_kernels_etile.hip) against the wave kernel and the torch oracle; timing at 2^20 walkers."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import sorted_walkers
import bench
from waveflow_amd.utils import physics
m, flat = bench.he_model("auto")
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
x = sorted_walkers(20000, 2, 9.0, 4)
def run(tile_min, xx):
    os.environ["WF_ENERGY_TILE_MIN"] = str(tile_min)
    h, p, l = m.hamiltonian(xx, protons, return_psi=True, return_laplacian=True)
    return [np.asarray(t.cpu() if hasattr(t, "cpu") else t, dtype=np.float64) for t in (h, p, l)]
wave = run(0, x)
tile = run(1, x)
for name, a, b in zip(("H psi", "psi", "laplacian"), tile, wave):
    sc = np.abs(b).max()
    d = np.abs(a - b)
    print(f"{name:10s} tile vs wave: max |diff| / max|.| = {d.max() / sc:.3e}   median {np.median(d) / sc:.3e}   finite {np.isfinite(a).all()}")
if os.environ.get("ORACLE", "1") == "1":
    from oracle import energy_torch as et
    mo = et.he_model(torch.float64)
    xs = x[:96]
    ho, po, lo = et.hamiltonian(mo, flat, xs.astype(np.float64), protons)
    for tag, tm in (("tile", 1), ("wave", 0)):
        h, p, l = run(tm, xs)
        print(f"{tag}: vs oracle  psi {np.abs(p - po).max() / np.abs(po).max():.2e}  lap {np.abs(l - lo).max() / np.abs(lo).max():.2e}  hpsi {np.abs(h - ho).max() / np.abs(ho).max():.2e}")
xb = torch.as_tensor(sorted_walkers(1 << 20, 2, 10.0, 1234)).cuda()
for tag, tm in (("tile", 1), ("wave", 0)):
    os.environ["WF_ENERGY_TILE_MIN"] = str(tm)
    for _ in range(3): m.hamiltonian(xb, protons)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10 if tm else 3
    for _ in range(n): m.hamiltonian(xb, protons)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{tag}: 2^20 walkers in {dt * 1e3:.3f} ms = {(1 << 20) / dt:.3e} walkers/s")
