import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import benchmark_tests
from waveflow_amd.vqmc import flatten_params
X = benchmark_tests.get_dataset("circles", 20000, 0.025, 0)
xd = torch.as_tensor(X.astype(np.float32)).cuda()
for mt in ("Flow", "IFlow", "MFlow"):
    params, log_pdf, sample = benchmark_tests.get_model(mt, 0.01, spline_degree=5, num_layers=3, num_knots=15)(1, 2)
    m = log_pdf.model
    x = torch.as_tensor(flatten_params(params).astype(np.float32)).cuda()
    mm, vv = torch.zeros_like(x), torch.zeros_like(x)
    ls = []
    for step in range(1, 301):
        m.set_params_device(x)
        lp, grad = m.logpdf_loss_grad(xd, -1.0 / xd.shape[0])
        s = m.block_sums(lp).cpu().tolist()
        ls.append(-s[0] / s[2])
        if step == 1:
            print(mt, "grad norm", float(grad.norm()), "n", x.numel())
        m.adam_step(x, grad, mm, vv, step, 1e-3)
    print(mt, "separate:", ls[0], ls[1], ls[-1])
    x2 = torch.as_tensor(flatten_params(params).astype(np.float32)).cuda()
    st = m.make_train_state(x2, torch.zeros_like(x2), torch.zeros_like(x2), 1, ring_len=512)
    m.set_params_device(x2)
    for step in range(1, 301):
        m.mle_train_step(st, xd, 1e-3)
    r = st["ring"].cpu().numpy()
    print(mt, "fused:   ", -r[1, 0] / r[1, 2], -r[2, 0] / r[2, 2], -r[300, 0] / r[300, 2], "max |dx|", float((x - x2).abs().max()))
