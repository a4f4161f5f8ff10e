import torch, time, sys, os
sys.path.insert(0, os.getcwd())
from waveflow_amd.flows import unconstrained_RQS
for K in (8, 32):
    N = 1 << 24 if K == 8 else 1 << 22
    uw = torch.randn(N, K, device="cuda"); uh = torch.randn(N, K, device="cuda"); ud = torch.randn(N, K - 1, device="cuda")
    x = torch.rand(N, device="cuda") * 2 - 1
    for _ in range(3): unconstrained_RQS(x, uw, uh, ud)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): unconstrained_RQS(x, uw, uh, ud)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    byts = N * ((3 * K - 1) * 4 + 4 + 8)   # K widths + K heights + K-1 derivatives + x in, y + logabsdet out
    print(f"K={K} N={N}: {dt*1e3:.3f} ms  {N/dt:.3e} elem/s  {byts/dt/1e9:.0f} GB/s algorithmic ({byts/dt/8e12*100:.1f}% of 8 TB/s)")
