import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["WF_POISON"] = os.environ.get("POISON", "1")
from waveflow_amd import model_factory
knots = int(os.environ.get("KNOTS", 33))
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=knots, n_i_internal_knots=knots, i_spline_reg=0.05,
                                            i_spline_reverse_fun_tol=1e-6, n_flow_layers=2, box_size=10, xu_coord_type="mean")
params, psi, log_pdf, sample = init_fun(3, 2)
m = psi.model
m.ensure_params(params)
os.environ["WF_WAVE_SAMPLE_MAX"] = "0"
os.environ["WF_SAMPLE_TILE_MIN"] = "0"
for B in (1000,):
    outs = []
    for i in range(3):
        x, lat = m.sample(8, B, return_latent=True, exact=True)
        torch.cuda.synchronize()
        outs.append((x.clone(), lat.clone()))
        print("knots", knots, "B", B, "call", i, "nan lat", torch.isnan(lat).sum(0).tolist(), "nan x", torch.isnan(x).sum().item(), "lat mean", torch.nanmean(lat, 0).tolist())
    print("   equal x:", torch.equal(outs[0][0], outs[1][0]), torch.equal(outs[1][0], outs[2][0]))
x, lat = m.sample(8, 40000, return_latent=True, exact=True)
idx = torch.isnan(lat[:, 1]).nonzero().flatten().cpu().numpy()
print("nan count", len(idx), "first 40:", idx[:40].tolist())
d = np.diff(idx)
print("run structure: fraction of consecutive indices", float((d == 1).mean()) if len(d) else 0.0, "| idx mod 128 histogram (16 bins):", np.histogram(idx % 128, bins=16, range=(0, 128))[0].tolist())
print("blocks (idx // 128) with NaNs:", len(np.unique(idx // 128)), "of", 40000 // 128)
x2, lat2 = m.sample(8, 40000, return_latent=True, exact=True)
a, b = lat[:, 1].cpu().numpy(), lat2[:, 1].cpu().numpy()
both = ~np.isnan(a) & ~np.isnan(b)
print("finite in both:", both.sum(), "equal there:", (a[both] == b[both]).mean())
