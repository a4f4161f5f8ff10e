"""Parity statistics shared by tests/ and __graft_entry__.smoke() (test infrastructure, like everything under oracle/: never imported by
waveflow_amd/).  One definition of the numbers that go into the GPU test log and the smoke tail:

  * pass rate at the north-star tolerance |v - fp64| <= 1e-5 |fp64| of the HIP result and of the fp32 oracle, on all walkers and on the
    well-conditioned subset (every logarithm argument > COND_MIN, |log_pdf| > LOGP_MIN: tests/test_gpu_parity.py explains the choice);
  * the DIRECT rate |HIP - oracle32| <= 1e-5 |oracle32| on the same two sets, and next to it the same rate for EXACT arithmetic in place of
    the HIP result, |fp64 - oracle32| <= 1e-5 |oracle32| ("direct_exact"): the fp32 oracle carries its own rounding errors, so this is what an
    evaluation with no error at all scores against it -- the ceiling for any kernel whose roundings are independent of the oracle's (only a
    kernel that repeats the oracle's operation order, like the scalar kernel, can score above it, by sharing its errors);
  * worst and median deviation from the fp64 build of the oracle.
"""
import numpy as np

COND_MIN, LOGP_MIN = 0.05, 1.0
RTOL = 1e-5


def stats(gpu, oracle32, truth, cond=None):
    gpu, oracle32, truth = (np.asarray(v, np.float64).reshape(-1) for v in (gpu, oracle32, truth))
    out = {"n": int(gpu.size)}

    def fill(tag, sel):
        g, o, t = gpu[sel], oracle32[sel], truth[sel]
        out[tag + "_n"] = int(g.size)
        if g.size == 0:
            return
        e_g, e_o = np.abs(g - t), np.abs(o - t)
        out[tag + "_pass_hip"] = float((e_g <= RTOL * np.abs(t)).mean())
        out[tag + "_pass_oracle32"] = float((e_o <= RTOL * np.abs(t)).mean())
        out[tag + "_direct"] = float((np.abs(g - o) <= RTOL * np.abs(o)).mean())
        out[tag + "_direct_exact"] = float((np.abs(t - o) <= RTOL * np.abs(o)).mean())
        out[tag + "_max_hip"], out[tag + "_max_oracle32"] = float(e_g.max()), float(e_o.max())
        out[tag + "_median_hip"], out[tag + "_median_oracle32"] = float(np.median(e_g)), float(np.median(e_o))

    fill("all", np.ones(gpu.size, bool))
    if cond is not None:
        fill("strict", (np.asarray(cond).reshape(-1) > COND_MIN) & (np.abs(truth) > LOGP_MIN))
    return out


def line(what, s):
    t = f"[parity {what}] n={s['n']}: 1e-5-relative pass rate vs fp64: HIP {s['all_pass_hip']:.5f} fp32-oracle {s['all_pass_oracle32']:.5f}; " \
        f"direct |HIP - oracle32| <= 1e-5 |oracle32|: {s['all_direct']:.5f} (exact arithmetic in place of HIP: {s['all_direct_exact']:.5f}); " \
        f"max |err| HIP {s['all_max_hip']:.2e} oracle {s['all_max_oracle32']:.2e}"
    if s.get("strict_n"):
        t += f" | well-conditioned subset ({s['strict_n']} walkers): HIP {s['strict_pass_hip']:.4f} fp32-oracle {s['strict_pass_oracle32']:.4f}, " \
             f"direct {s['strict_direct']:.4f} (exact arithmetic: {s['strict_direct_exact']:.4f}), max |err| HIP {s['strict_max_hip']:.2e} oracle {s['strict_max_oracle32']:.2e}"
    return t
