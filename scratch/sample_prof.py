"""rocprofv3 / timing target: the large-batch sampler and the inverse alone (2^17 walkers, shipped He model)."""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
import bench
m, flat = bench.he_model("auto")
B = int(os.environ.get("B", 1 << 17))
x, lat = m.sample(3, B, return_latent=True, exact=True)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
print("sample  :", t(lambda: m.sample(5, B, exact=True)) * 1e3, "ms")
print("inverse :", t(lambda: m.inverse(lat, exact=True)) * 1e3, "ms")
print("log_pdf :", t(lambda: m.log_pdf(x)) * 1e3, "ms")
