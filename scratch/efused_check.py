"""H psi of the He checkpoint through the one-kernel path (k_efused), the launch-per-net tile path (WF_ENERGY_FUSED=0) and the wave kernel
(WF_ENERGY_TILE_MIN=0): errors against the fp64 torch oracle next to the fp32 torch oracle's own, mutual differences on 70 001 walkers,
HIP-event time at 2^20 walkers."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import sorted_walkers
import bench
from oracle import energy_torch as et
from waveflow_amd.utils import physics
m, flat = bench.he_model("auto")
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
g = np.load("tests/golden/he_golden.npz")
x = np.concatenate([np.sort(g["sample_points"], -1), sorted_walkers(250, 2, 10.0, 5)]).astype(np.float32)


def run(xs, mode):
    env = {"fused": {"WF_ENERGY_TILE_MIN": "1"}, "pernet": {"WF_ENERGY_TILE_MIN": "1", "WF_ENERGY_FUSED": "0"}, "wave": {"WF_ENERGY_TILE_MIN": "0"}}[mode]
    os.environ.update(env)
    out = [np.asarray(t.cpu() if hasattr(t, "cpu") else t, dtype=np.float64) for t in m.hamiltonian(xs, protons, return_psi=True, return_laplacian=True)]
    for k in env:
        del os.environ[k]
    return out


ho64, po64, lo64 = et.hamiltonian(et.he_model(torch.float64), flat, x.astype(np.float64), protons)
ho32, po32, lo32 = et.hamiltonian(et.he_model(torch.float32), flat, x, protons)
print("fp32 torch oracle vs fp64: psi max %.2e  lap max %.2e median %.2e  hpsi max %.2e   (scale lap %.2e hpsi %.2e)" % (
    np.abs(po32 - po64).max(), np.abs(lo32 - lo64).max(), np.median(np.abs(lo32 - lo64)), np.abs(ho32 - ho64).max(), np.abs(lo64).max(), np.abs(ho64).max()))
for mode in ("fused", "pernet", "wave"):
    hp, ps, lap = run(x, mode)
    print("%-7s vs fp64: psi max %.2e  lap max %.2e (%.2f x oracle32) median %.2e (%.2f x)  hpsi max %.2e (%.2f x)" % (
        mode, np.abs(ps - po64).max(), np.abs(lap - lo64).max(), np.abs(lap - lo64).max() / np.abs(lo32 - lo64).max(), np.median(np.abs(lap - lo64)),
        np.median(np.abs(lap - lo64)) / np.median(np.abs(lo32 - lo64)), np.abs(hp - ho64).max(), np.abs(hp - ho64).max() / np.abs(ho32 - ho64).max()))
xb = sorted_walkers(70001, 2, 10.0, 21)
res = {mode: run(xb, mode) for mode in ("fused", "pernet", "wave")}
for a, b in (("fused", "wave"), ("pernet", "wave"), ("fused", "pernet")):
    for k, nm in enumerate(("hpsi", "psi", "lap")):
        d = np.abs(res[a][k] - res[b][k]); sc = np.abs(res[b][k]).max()
        print("%s vs %s %-4s: max %.2e median %.2e of the batch maximum" % (a, b, nm, d.max() / sc, np.median(d) / sc))
xt = torch.as_tensor(sorted_walkers(1 << 20, 2, 10.0, 1234)).cuda()
for mode in ("fused", "pernet"):
    env = {"fused": {"WF_ENERGY_TILE_MIN": "1"}, "pernet": {"WF_ENERGY_TILE_MIN": "1", "WF_ENERGY_FUSED": "0"}}[mode]
    os.environ.update(env)
    first = m.hamiltonian(xt, protons).clone()
    same = all(torch.equal(m.hamiltonian(xt, protons), first) for _ in range(5))
    ms = bench.event_ms(lambda: m.hamiltonian(xt, protons), 30, 20)
    print("%-7s 2^20 walkers: %.4f ms per call = %.3e walkers/s   bit-reproducible: %s" % (mode, ms, (1 << 20) / ms * 1e3, same))
    for k in env:
        del os.environ[k]

# BASELINE's "32-bin" variant (33 knots: two 32-row blocks per dimension), seeded parameters: the one-kernel form against the wave kernel
m33 = bench.seeded_model(2, 33, "auto")
xs = sorted_walkers(50001, 2, 10.0, 3)
os.environ["WF_ENERGY_TILE_MIN"] = "1"
a = [np.asarray(t, dtype=np.float64) for t in m33.hamiltonian(xs, protons, return_psi=True, return_laplacian=True)]
os.environ["WF_ENERGY_TILE_MIN"] = "0"
b = [np.asarray(t, dtype=np.float64) for t in m33.hamiltonian(xs, protons, return_psi=True, return_laplacian=True)]
for k, nm in enumerate(("hpsi", "psi", "lap")):
    d = np.abs(a[k] - b[k]); sc = np.abs(b[k]).max()
    print("33 knots: fused vs wave %-4s: max %.2e median %.2e of the batch maximum" % (nm, d.max() / sc, np.median(d) / sc))
for mode, tm in (("fused", "1"), ("wave", "0")):
    os.environ["WF_ENERGY_TILE_MIN"] = tm
    ms = bench.event_ms(lambda: m33.hamiltonian(xt, protons), 10 if mode == "fused" else 3, 5 if mode == "fused" else 1)
    print("33 knots %-5s 2^20 walkers: %.4f ms per call = %.3e walkers/s" % (mode, ms, (1 << 20) / ms * 1e3))
del os.environ["WF_ENERGY_TILE_MIN"]
