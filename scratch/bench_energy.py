import torch, time, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import bench
m, flat = bench.he_model("auto")
for B in (1 << 14, 1 << 18):
    x = bench.walkers(B, 1).cuda()
    for _ in range(2): m.hamiltonian(x, [0.0, 0.0])
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): h = m.hamiltonian(x, [0.0, 0.0])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"hamiltonian B={B}: {dt*1e3:.2f} ms  {B/dt:.3e} walkers/s")
# sampler + local energy = one VQMC energy estimate (exact inverse so that samples follow |psi|^2)
xs = m.sample(7, 1 << 16, exact=True)
h, ps = m.hamiltonian(xs, [0.0, 0.0], return_psi=True)
el = (h / (ps + 1e-8)).double()
print("VQMC <E_L> over 65536 |psi|^2 samples: mean %.3f  median %.3f  std %.1f" % (el.mean().item(), el.median().item(), el.std().item()))
t = time.perf_counter(); xs = m.sample(8, 1 << 18, exact=True); torch.cuda.synchronize(); print("sample 2^18: %.2f ms" % ((time.perf_counter() - t) * 1e3))
