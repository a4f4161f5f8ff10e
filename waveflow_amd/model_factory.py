"""waveflow.model_factory call surface (reference: model_factory.py:7-146), same names and keyword arguments.

The reference assembles closures; here the factories only assemble *descriptions* (flows.* dataclasses) which
core.DeviceModel turns into one wf_model_desc for the HIP library.
"""
import numpy as np

from . import flows, wavefunctions
from .flows import MaskedTransform


def get_masked_transform(return_simple_masked_transform=False, allow_negative_params=False):
    """model_factory.py:7-93: returns the conditioner description consumed by IMADE / MADE / Waveflow / MFlow."""
    return MaskedTransform(simple=bool(return_simple_masked_transform), allow_negative_params=bool(allow_negative_params))


def _imade_stack(depth, degree, knots, reg, tol, left, right, zero_grad=False, head=()):
    """`depth` x (IMADE, Reverse) behind the optional `head` layers -> flows.Serial."""
    spec = dict(spline_degree=degree, n_internal_knots=knots, spline_regularization=reg, reverse_fun_tol=tol,
                constraints_dict_left=left, constraints_dict_right=right, set_nn_output_grad_to_zero=zero_grad)
    chain = list(head)
    for _ in range(depth):
        chain.append(flows.IMADE(get_masked_transform(), **spec))
        chain.append(flows.Reverse())
    return flows.Serial(*chain)


def get_model(base_spline_degree=5, i_spline_degree=5, n_prior_internal_knots=15, n_i_internal_knots=15,
              i_spline_reg=0, i_spline_reverse_fun_tol=0.000001, n_flow_layers=1,
              prior_constraint_dict_left={}, prior_constraint_dict_right={}, i_constraint_dict_left={}, i_constraint_dict_right={},
              set_nn_output_grad_to_zero=False):
    """model_factory.py:96-116: density model = IMADE stack + autoregressive M-spline prior."""
    bijection = _imade_stack(n_flow_layers, i_spline_degree, n_i_internal_knots, i_spline_reg, i_spline_reverse_fun_tol,
                             i_constraint_dict_left, i_constraint_dict_right, zero_grad=set_nn_output_grad_to_zero)
    prior = dict(spline_degree=base_spline_degree, n_internal_knots=n_prior_internal_knots,
                 constraints_dict_left=prior_constraint_dict_left, constraints_dict_right=prior_constraint_dict_right,
                 set_nn_output_grad_to_zero=set_nn_output_grad_to_zero)
    return flows.MFlow(bijection, get_masked_transform(), **prior)


def get_waveflow_model(n_dimension, base_spline_degree=5, i_spline_degree=5, n_prior_internal_knots=16, n_i_internal_knots=16,
                       i_spline_reg=0, i_spline_reverse_fun_tol=0.000001,
                       n_flow_layers=1, box_size=1, xu_coord_type='mean'):
    """model_factory.py:121-146: wavefunction model = box transform + IMADE stack + orthogonal-B-spline prior."""
    n_dimension = int(n_dimension)
    # the antisymmetry node sits on every unit-cube coordinate except the centre-of-mass one ('mean': last, else first)
    nodes = np.arange(n_dimension - 1, dtype=int) + (0 if xu_coord_type == 'mean' else 1)
    bijection = _imade_stack(n_flow_layers, i_spline_degree, n_i_internal_knots, i_spline_reg, i_spline_reverse_fun_tol,
                             {0: 0}, {0: 1}, head=[flows.BoxTransformLayer(box_size, xu_coord_type=xu_coord_type)])
    prior = dict(spline_degree=base_spline_degree, n_internal_knots=n_prior_internal_knots,
                 constraints_dict_left={0: 0}, constraints_dict_right={0: 0},
                 constrained_dimension_indices_left=nodes, constrained_dimension_indices_right=np.zeros(0, dtype=int),
                 set_nn_output_grad_to_zero=False)
    return wavefunctions.Waveflow(bijection, get_masked_transform(allow_negative_params=True), **prior)
