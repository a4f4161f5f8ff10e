import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import sorted_walkers
import oracle
from waveflow_amd import flows, model_factory, wavefunctions, flatten_params
mt = model_factory.get_masked_transform
cases = {
  "both": ({0: 0.0, 1: 0.0}, {0: 1.0, 1: 0.0}, {0: 0, 2: 0, 3: 0}, {0: 0, 1: 0}),
  "I only": ({0: 0.0, 1: 0.0}, {0: 1.0, 1: 0.0}, {0: 0}, {0: 0}),
  "prior only": ({0: 0.0}, {0: 1.0}, {0: 0, 2: 0, 3: 0}, {0: 0, 1: 0}),
  "zero only": ({0: 0.0}, {0: 1.0}, {0: 0}, {0: 0}),
}
for REG in (0.01, 0.05):
  print('reg', REG)
  for name, (il, ir, pl, pr) in cases.items():
      init = wavefunctions.Waveflow(
          flows.Serial(flows.BoxTransformLayer(2.0), flows.IMADE(mt(), 5, 16, REG, 1e-6, il, ir), flows.Reverse()),
          mt(allow_negative_params=True), 5, 16, constraints_dict_left=pl, constraints_dict_right=pr,
          constrained_dimension_indices_left=[0], set_nn_output_grad_to_zero=False)
      params, psi, log_pdf, _ = init(5, 2)
      om = oracle.Model(D=2, n_layers=1, box="mean", box_L=2.0, i_k=5, i_knots=16, i_reg=REG, i_left=il, i_right=ir, prior="waveflow", p_k=5,
                        p_knots=16, p_left=pl, p_right=pr, constr_left=(0,))
      x = sorted_walkers(20000, 2, 2.0, 9)
      flat = flatten_params(params)
      t = om.log_pdf(flat, x, f64=True); o32 = om.log_pdf(flat, x)
      u64 = om.log_pdf(flat, x, return_u=True, f64=True)[1]
      for k in ("scalar", "mfma", "wave"):
          log_pdf.model.set_kernel(k)
          lp, u = log_pdf(params, x, return_sample=True)
          e = np.abs(np.asarray(lp) - t)
          tol = 2e-5 + 1e-5 * np.abs(t)
          print("   outside tol: HIP", int((e > tol).sum()), "oracle32", int((np.abs(o32 - t) > tol).sum()))
          print(f"{name:11s} {k:7s} log_pdf err: median {np.median(e):.2e} p99 {np.quantile(e, .99):.2e} max {e.max():.2e}   u err max {np.abs(np.asarray(u) - u64).max():.2e}"
                f"   | oracle32: median {np.median(np.abs(o32 - t)):.2e} p99 {np.quantile(np.abs(o32 - t), .99):.2e} max {np.abs(o32 - t).max():.2e}")
