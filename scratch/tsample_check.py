"""The staged sampler / inverse of large two-particle batches against the one-walker-per-wave kernel: inverse on the same latent points,
round trip through the forward pass, moments of the draws, timings.   usage: [B=131072] python3 scratch/tsample_check.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
import bench
m, flat = bench.he_model("auto")
B = int(os.environ.get("B", 1 << 17))
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for exact in (True, False):
    os.environ["WF_SAMPLE_TILE_MIN"] = "0"
    xw, lw = m.sample(7, B, return_latent=True, exact=exact)
    tw = t(lambda: m.sample(7, B, exact=exact))
    iw = m.inverse(lw, exact=exact)
    tiw = t(lambda: m.inverse(lw, exact=exact))
    os.environ["WF_SAMPLE_TILE_MIN"] = "16384"
    xt, lt = m.sample(7, B, return_latent=True, exact=exact)
    tt = t(lambda: m.sample(7, B, exact=exact))
    it = m.inverse(lw, exact=exact)          # same latent points as the wave kernel inverted
    tit = t(lambda: m.inverse(lw, exact=exact))
    d = (it - iw).abs()
    print(f"exact={exact}: inverse staged vs wave on the same latents: max |dx| {d.max().item():.3e} median {d.median().item():.3e} "
          f"frac > 1e-4: {(d > 1e-4).float().mean().item():.2e}  finite {torch.isfinite(it).all().item()}")
    print(f"   time: sample wave {tw * 1e3:.3f} ms staged {tt * 1e3:.3f} ms | inverse wave {tiw * 1e3:.3f} ms staged {tit * 1e3:.3f} ms")
    # latent column 1 uses the same stream and bound as the wave kernel: mostly the same draws; column 0 has the tighter bound: another draw of the same law
    same1 = ((lt[:, 1] - lw[:, 1]).abs() < 1e-6).float().mean().item()
    print(f"   latent: column 1 equal to the wave kernel's for {same1:.4f} of the walkers (given a different column 0: expected ~0); NaNs {torch.isnan(xt).sum().item()}")
    for name, a, b2 in (("latent0", lt[:, 0], lw[:, 0]), ("latent1", lt[:, 1], lw[:, 1]), ("x0", xt[:, 0], xw[:, 0]), ("x1", xt[:, 1], xw[:, 1])):
        print(f"   {name}: mean {a.mean().item():+.5f} / {b2.mean().item():+.5f}  std {a.std().item():.5f} / {b2.std().item():.5f}  (staged / wave; MC error ~{a.std().item() / np.sqrt(B):.1e})")
    if exact:
        # round trip: forward of the staged samples -> latent (the model's flow_fwd), should return the drawn latent
        lp_t = m.log_pdf(xt); lp_w = m.log_pdf(xw)
        print(f"   mean log_pdf of the draws: staged {lp_t.mean().item():.5f} wave {lp_w.mean().item():.5f}  (MC error ~{lp_w.std().item() / np.sqrt(B):.1e})")
