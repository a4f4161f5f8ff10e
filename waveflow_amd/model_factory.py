"""waveflow.model_factory call surface (reference: model_factory.py:7-146), same names and keyword arguments."""
import numpy as np

from . import flows, wavefunctions
from .flows import MaskedTransform


def get_masked_transform(return_simple_masked_transform=False, allow_negative_params=False):
    """model_factory.py:7-93: returns the conditioner description consumed by IMADE / MADE / Waveflow / MFlow."""
    return MaskedTransform(simple=bool(return_simple_masked_transform), allow_negative_params=bool(allow_negative_params))


def get_model(base_spline_degree=5, i_spline_degree=5, n_prior_internal_knots=15, n_i_internal_knots=15,
              i_spline_reg=0, i_spline_reverse_fun_tol=0.000001, n_flow_layers=1,
              prior_constraint_dict_left={}, prior_constraint_dict_right={}, i_constraint_dict_left={}, i_constraint_dict_right={},
              set_nn_output_grad_to_zero=False):
    """model_factory.py:96-116"""
    layers = []
    for _ in range(n_flow_layers):
        layers += [flows.IMADE(get_masked_transform(), spline_degree=i_spline_degree, n_internal_knots=n_i_internal_knots,
                               spline_regularization=i_spline_reg, reverse_fun_tol=i_spline_reverse_fun_tol,
                               constraints_dict_left=i_constraint_dict_left, constraints_dict_right=i_constraint_dict_right,
                               set_nn_output_grad_to_zero=set_nn_output_grad_to_zero),
                   flows.Reverse()]
    return flows.MFlow(flows.Serial(*layers), get_masked_transform(),
                       spline_degree=base_spline_degree, n_internal_knots=n_prior_internal_knots,
                       constraints_dict_left=prior_constraint_dict_left, constraints_dict_right=prior_constraint_dict_right,
                       set_nn_output_grad_to_zero=set_nn_output_grad_to_zero)


def get_waveflow_model(n_dimension, base_spline_degree=5, i_spline_degree=5, n_prior_internal_knots=16, n_i_internal_knots=16,
                       i_spline_reg=0, i_spline_reverse_fun_tol=0.000001,
                       n_flow_layers=1, box_size=1, xu_coord_type='mean'):
    """model_factory.py:121-146"""
    if xu_coord_type == 'mean':
        constrained_left = np.arange(0, n_dimension - 1, dtype=int)
    else:
        constrained_left = np.arange(1, n_dimension, dtype=int)
    layers = [flows.BoxTransformLayer(box_size, xu_coord_type=xu_coord_type)]
    for _ in range(n_flow_layers):
        layers += [flows.IMADE(get_masked_transform(), spline_degree=i_spline_degree, n_internal_knots=n_i_internal_knots,
                               spline_regularization=i_spline_reg, reverse_fun_tol=i_spline_reverse_fun_tol,
                               constraints_dict_left={0: 0}, constraints_dict_right={0: 1},
                               set_nn_output_grad_to_zero=False),
                   flows.Reverse()]
    return wavefunctions.Waveflow(flows.Serial(*layers), get_masked_transform(allow_negative_params=True),
                                  spline_degree=base_spline_degree, n_internal_knots=n_prior_internal_knots,
                                  constraints_dict_left={0: 0}, constraints_dict_right={0: 0},
                                  constrained_dimension_indices_left=constrained_left,
                                  constrained_dimension_indices_right=np.array([], dtype=int),
                                  set_nn_output_grad_to_zero=False)
