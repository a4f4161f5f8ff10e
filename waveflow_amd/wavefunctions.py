"""waveflow.wavefunctions call surface on the HIP path (reference: wavefunctions.py:9-112)."""
import numpy as np

from . import _lib
from .flows import _InitFun, as_generator, seed_from


class Waveflow(_InitFun):
    """init_fun(rng, input_dim) -> (params, psi, log_pdf, sample)   (wavefunctions.py:110)"""

    def __init__(self, transformation, sp_transformation, spline_degree, n_internal_knots, constraints_dict_left={0: 0, 2: 0},
                 constraints_dict_right={0: 0}, constrained_dimension_indices_left=(), constrained_dimension_indices_right=(),
                 set_nn_output_grad_to_zero=True, n_spline_base_mesh_points=2000):
        self.gate = bool(set_nn_output_grad_to_zero)   # (get_waveflow_model passes False, model_factory.py:143)
        if sp_transformation.simple or not sp_transformation.allow_negative_params:
            raise NotImplementedError("Waveflow needs get_masked_transform(allow_negative_params=True) for the prior head")
        if len(np.asarray(constrained_dimension_indices_right).reshape(-1)):
            raise NotImplementedError("constrained_dimension_indices_right is unused by the reference (wavefunctions.py:48,67)")
        self.transformation, self.sp = transformation, sp_transformation
        self.k, self.n = spline_degree, n_internal_knots
        self.left, self.right = dict(constraints_dict_left), dict(constraints_dict_right)
        self.constrained_left = [int(c) for c in np.asarray(constrained_dimension_indices_left).reshape(-1)]
        self.n_mesh = n_spline_base_mesh_points

    def __call__(self, rng, input_dim):
        g = as_generator(rng)
        tparams = self.transformation.init_params(g, input_dim)
        nb = self.n + self.k - 1  # bsplines_jax.py:58-65: n_knots - k - 1 with (k+1)-fold end knots
        sparams = self.sp.init_params(g, input_dim, nb)
        model = self.transformation.fused_model(input_dim, prior=_lib.PRIOR_WAVEFLOW, p_degree=self.k, p_knots=self.n,
                                                p_left=self.left, p_right=self.right,
                                                constrained_left=self.constrained_left, n_mesh=self.n_mesh, p_gate=self.gate)
        if model is None:
            raise NotImplementedError("Waveflow: the bijector stack must be [Box] + (IMADE, Reverse)*n")
        assert model.p_nb == nb

        def log_pdf(params, inputs, return_sample=False):
            model.ensure_params(params)
            return model.log_pdf(inputs, return_sample=return_sample)

        def psi(params, inputs, log_tol=1e-7):
            model.ensure_params(params)
            return model.psi(inputs)

        def sample(rng, params, num_samples=1, return_original_samples=False, exact_inverse=False):
            """wavefunctions.py:74-107.  exact_inverse=False keeps the reference's IMADE.inverse_fun (made.py:88): the
            conditioner sees the values being inverted, so the samples follow a distorted density; True draws from |psi|^2."""
            model.ensure_params(params)
            return model.sample(seed_from(rng), num_samples, return_latent=return_original_samples, exact=exact_inverse)

        log_pdf.model = psi.model = sample.model = model
        return (tparams, sparams), psi, log_pdf, sample
