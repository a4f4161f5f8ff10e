"""waveflow.flows call surface on the HIP path.

Same protocol as the reference (flows/bijections/bijections.py:7-16):
    init_fun(rng, input_dim, **kw) -> (params, direct_fun, inverse_fun)
    direct_fun(params, inputs[B, D]) -> (outputs[B, D], log_det_jacobian[B])
and for distributions (flows/distributions.py):
    init_fun(rng, input_dim) -> (params, log_pdf, sample)

Every constructor returns a callable init_fun that also carries a `.spec`, so that containers
(Serial, Flow, MFlow, Waveflow) can fuse the whole stack into one wf_model / one kernel launch.
`rng` may be an int seed, a numpy Generator or None: initial parameters follow the reference's
distributions (model_factory.py:25-28, 84) but not JAX's threefry stream (parity unpinned).
"""
from dataclasses import dataclass, field

import numpy as np

from .. import _lib
from ..core import DeviceModel, flatten_params
from .neural_splines import FCNN, RQS, NeuralSplineCoupling, unconstrained_RQS  # noqa: F401

HIDDEN = 64  # model_factory.py:72


def as_generator(rng):
    if isinstance(rng, np.random.Generator):
        return rng
    if rng is None:
        return np.random.default_rng(0)
    if isinstance(rng, (int, np.integer)):
        return np.random.default_rng(int(rng))
    # e.g. a JAX-style key array: hash its bytes into a seed
    return np.random.default_rng(int(np.asarray(rng).astype(np.uint64).sum()) & 0x7FFFFFFF)


def seed_from(rng):
    """A 63-bit seed for the device sampler from whatever the caller uses as `rng`."""
    return int(as_generator(rng).integers(0, 2 ** 63 - 1))


# --------------------------------------------------------------------------- conditioner
@dataclass
class MaskedTransform:
    """model_factory.get_masked_transform(...) result (model_factory.py:7-93)."""
    simple: bool = False
    allow_negative_params: bool = False

    def init_params(self, rng, input_dim, output_shape=2):
        """MaskedDense x3 init (model_factory.py:22-29) + zero_params (:84)."""
        g = as_generator(rng)
        D, H = input_dim, HIDDEN
        layers = []
        for fan_in, n_out in ((D, H), (H, H), (H, D * output_shape)):
            bound = 1.0 / np.sqrt(fan_in)
            W = g.uniform(-bound, bound, size=(fan_in, n_out)).astype(np.float32)
            b = g.uniform(-bound, bound, size=(n_out,)).astype(np.float32)
            layers.append((W, b))
        nn_params = [layers[0], (), layers[1], (), layers[2]]  # stax.serial(Dense, Tanh, Dense, Tanh, Dense)
        if self.simple:
            return nn_params
        zero_params = g.uniform(-0.5, 0.5, size=(D, output_shape)).astype(np.float32)
        return (nn_params, zero_params)


# --------------------------------------------------------------------------- specs
@dataclass
class IMADESpec:
    degree: int
    knots: int
    reg: float
    tol: float
    left: dict
    right: dict
    n_mesh: int = 2000
    gate: bool = False     # set_nn_output_grad_to_zero (model_factory.py:64-67)


@dataclass
class MADESpec:
    pass


@dataclass
class NSCSpec:
    """NeuralSplineCoupling(K, B, hidden_dim) (neural_splines.py:244)"""
    K: int
    B: float
    hidden: int


@dataclass
class BoxSpec:
    box_side: float
    kind: str


@dataclass
class ReverseSpec:
    pass


@dataclass
class SerialSpec:
    items: list = field(default_factory=list)


def _desc(D, layers=(), box=None, prior=_lib.PRIOR_UNIFORM, p_degree=0, p_knots=0, p_left=None, p_right=None,
          normal_offset=0.0, constrained_left=(), n_mesh=None, p_gate=False):
    """Fill a wf_model_desc from specs.  `layers`: list of identical IMADESpec or MADESpec.  `n_mesh`: the prior spline's mesh
    (n_spline_base_mesh_points); wf_model_desc carries ONE mesh size for the layers' and the prior's tables."""
    d = _lib.ModelDesc()
    d.n_dim, d.hidden, d.n_flow_layers, d.n_mesh = D, HIDDEN, len(layers), (2000 if n_mesh is None else int(n_mesh))
    d.layer_kind = _lib.LAYER_IMADE
    if layers:
        first = layers[0]
        if any(l != first for l in layers):
            raise NotImplementedError("the HIP path fuses stacks of identically configured layers only")
        if isinstance(first, IMADESpec):
            d.layer_kind = _lib.LAYER_IMADE
            d.i_degree, d.i_knots, d.i_reg = first.degree, first.knots, first.reg
            d.i_left, d.i_right = _lib.BC.from_dict(first.left), _lib.BC.from_dict(first.right)
            if n_mesh is not None and int(n_mesh) != int(first.n_mesh) and prior in (_lib.PRIOR_WAVEFLOW, _lib.PRIOR_MFLOW):
                raise NotImplementedError(f"prior spline mesh ({n_mesh} points) != flow-layer spline mesh ({first.n_mesh}): the fused "
                                          "model keeps one mesh size; build the two with equal n_spline_base_mesh_points")
            d.n_mesh = first.n_mesh
            d.i_reverse_tol = float(first.tol) if first.tol is not None else 0.0
            d.i_gate = int(bool(first.gate))
        else:
            d.layer_kind = _lib.LAYER_MADE
    if box is not None:
        d.box_kind = {"mean": _lib.BOX_MEAN, "first": _lib.BOX_FIRST}[box.kind if box.kind == "mean" else "first"]
        d.box_size = float(box.box_side)
    d.prior_kind = prior
    d.p_degree, d.p_knots = p_degree, p_knots
    d.p_left, d.p_right = _lib.BC.from_dict(p_left), _lib.BC.from_dict(p_right)
    d.normal_offset = normal_offset
    d.p_gate = int(bool(p_gate))
    cl = [int(c) for c in np.asarray(constrained_left).reshape(-1)]
    d.n_constrained_left = len(cl)
    for i, c in enumerate(cl):
        d.constrained_left[i] = c
    return d


def parse_nsc_serial(spec):
    """SerialSpec -> (NSCSpec, n_layers, reverse) if it is NeuralSplineCoupling * n or (NeuralSplineCoupling, Reverse) * n, else None."""
    items = list(spec.items)
    if not items or not isinstance(items[0], NSCSpec):
        return None
    reverse = len(items) > 1 and isinstance(items[1], ReverseSpec)
    step = 2 if reverse else 1
    if len(items) % step:
        return None
    for i, it in enumerate(items):
        want_rev = reverse and i % 2 == 1
        if want_rev != isinstance(it, ReverseSpec) or (not want_rev and it != items[0]):
            return None
    return items[0], len(items) // step, reverse


def parse_serial(spec):
    """SerialSpec -> (box or None, [layer specs]) if it is [Box]? + (IMADE|MADE, Reverse)*n, else None."""
    items = list(spec.items)
    box = None
    if items and isinstance(items[0], BoxSpec):
        box = items.pop(0)
    if len(items) % 2:
        return None
    layers = []
    for a, b in zip(items[0::2], items[1::2]):
        if not isinstance(a, (IMADESpec, MADESpec)) or not isinstance(b, ReverseSpec):
            return None
        layers.append(a)
    if layers and any(l != layers[0] for l in layers):
        return None
    return box, layers


def _flip(a):
    return a.flip(-1) if hasattr(a, "flip") else np.ascontiguousarray(np.asarray(a)[..., ::-1])


class _InitFun:
    """A callable init_fun carrying `.spec`."""
    spec = None

    def __call__(self, rng, input_dim, **kwargs):
        raise NotImplementedError


# --------------------------------------------------------------------------- bijectors
class IMADE(_InitFun):
    """flows/bijections/made.py:44-105"""

    def __init__(self, transform, spline_degree=4, n_internal_knots=12, spline_regularization=0.0, reverse_fun_tol=0.0001,
                 constraints_dict_left={0: 0.0}, constraints_dict_right={0: 1.0}, set_nn_output_grad_to_zero=False,
                 n_spline_base_mesh_points=2000):
        if transform.simple or transform.allow_negative_params:
            raise NotImplementedError("IMADE needs get_masked_transform() (sigmoid head), as in the reference's factories")
        self.transform = transform
        self.spec = IMADESpec(spline_degree, n_internal_knots, float(spline_regularization), reverse_fun_tol,
                              dict(constraints_dict_left), dict(constraints_dict_right), n_spline_base_mesh_points,
                              bool(set_nn_output_grad_to_zero))

    def n_bases(self):
        return self.spec.knots + self.spec.degree  # knots + 2*(k+1) - 2 - k, isplines_jax.py:91-95

    def init_params(self, rng, input_dim):
        return self.transform.init_params(rng, input_dim, self.n_bases())

    def __call__(self, rng, input_dim, **kwargs):
        params = self.init_params(rng, input_dim)
        model = DeviceModel(_desc(input_dim, [self.spec]))

        def direct_fun(params, inputs, **kw):
            model.ensure_params(params)
            return model.layer(0, inputs)

        def inverse_fun(params, inputs, exact=False, **kw):
            model.ensure_params(params)
            # the fused model inverts (layer, Reverse); a stand-alone layer has no Reverse, so undo it on the way in
            return model.inverse(_flip(inputs), exact=exact), 0   # the reference returns 0 as log-det here (made.py:100)

        return params, direct_fun, inverse_fun


class MADE(_InitFun):
    """flows/bijections/made.py:7-41 with model_factory.simple_masked_transform"""

    def __init__(self, transform):
        if not transform.simple:
            raise NotImplementedError("MADE needs get_masked_transform(return_simple_masked_transform=True)")
        self.transform = transform
        self.spec = MADESpec()

    def init_params(self, rng, input_dim):
        return self.transform.init_params(rng, input_dim, 2)

    def __call__(self, rng, input_dim, **kwargs):
        params = self.init_params(rng, input_dim)
        model = DeviceModel(_desc(input_dim, [self.spec]))

        def direct_fun(params, inputs, **kw):
            model.ensure_params(params)
            return model.layer(0, inputs)

        def inverse_fun(params, inputs, **kw):
            model.ensure_params(params)
            return model.inverse(_flip(inputs)), 0

        return params, direct_fun, inverse_fun


class BoxTransformLayer(_InitFun):
    """flows/bijections/made.py:108-204"""

    def __init__(self, box_side=1, xu_coord_type="mean"):
        self.spec = BoxSpec(float(box_side), "mean" if xu_coord_type == "mean" else "first")

    def init_params(self, rng, input_dim):
        return ()

    def __call__(self, rng, input_dim, **kwargs):
        model = DeviceModel(_desc(input_dim, [], box=self.spec))
        model.set_params(np.zeros(0, np.float32))

        def direct_fun(params, inputs, **kw):
            return model.flow(inputs)

        def reverse_fun(params, inputs, **kw):
            return model.inverse(inputs), 0

        return (), direct_fun, reverse_fun


class Reverse(_InitFun):
    """flows/bijections/bijections.py:317-347: column reversal, log-det 0."""

    def __init__(self):
        self.spec = ReverseSpec()

    def init_params(self, rng, input_dim):
        return ()

    def __call__(self, rng, input_dim, **kwargs):
        def direct_fun(params, inputs, **kw):
            if hasattr(inputs, "flip"):
                return inputs.flip(-1), inputs.new_zeros(inputs.shape[:1])
            inputs = np.asarray(inputs)
            return inputs[:, ::-1].copy(), np.zeros(inputs.shape[:1], inputs.dtype)

        return (), direct_fun, direct_fun


class Serial(_InitFun):
    """flows/bijections/bijections.py:417-467"""

    def __init__(self, *init_funs):
        self.init_funs = list(init_funs)
        self.spec = SerialSpec([f.spec for f in init_funs])

    def init_params(self, rng, input_dim):
        g = as_generator(rng)
        return [f.init_params(g, input_dim) for f in self.init_funs]

    def fused_model(self, input_dim, **prior_kw):
        nsc = parse_nsc_serial(self.spec)
        if nsc is not None:   # the coupling stack as a model (wf_model_desc.layer_kind = WF_LAYER_NSC): Normal / Uniform priors
            first, n, reverse = nsc
            if prior_kw.get("prior", _lib.PRIOR_UNIFORM) not in (_lib.PRIOR_NORMAL, _lib.PRIOR_UNIFORM):
                raise NotImplementedError("NeuralSplineCoupling stacks are built under flows.Flow(.., Normal() | Uniform())")
            d = _lib.ModelDesc()
            d.n_dim, d.hidden, d.n_flow_layers, d.n_mesh, d.layer_kind = input_dim, HIDDEN, n, 2000, _lib.LAYER_NSC
            d.prior_kind = prior_kw.get("prior", _lib.PRIOR_UNIFORM)
            d.normal_offset = float(prior_kw.get("normal_offset", 0.0))
            d.nsc_bins, d.nsc_tail_bound, d.nsc_hidden, d.nsc_reverse = first.K, first.B, first.hidden, int(reverse)
            return DeviceModel(d)
        parsed = parse_serial(self.spec)
        if parsed is None:
            return None
        box, layers = parsed
        return DeviceModel(_desc(input_dim, layers, box=box, **prior_kw))

    def __call__(self, rng, input_dim, **kwargs):
        params = self.init_params(rng, input_dim)
        model = self.fused_model(input_dim)
        if model is not None:
            def direct_fun(params, inputs, **kw):
                model.ensure_params(params)
                return model.flow(inputs)

            def inverse_fun(params, inputs, exact=False, **kw):
                model.ensure_params(params)
                return model.inverse(inputs, exact=exact), 0
            return params, direct_fun, inverse_fun
        # general composition: one launch per layer (still the HIP path)
        subs = [f(rng, input_dim) for f in self.init_funs]

        def direct_fun(params, inputs, **kw):
            total = None
            for (_, dfun, _), p in zip(subs, params):
                inputs, ld = dfun(p, inputs)
                total = ld if total is None else total + ld
            return inputs, total

        def inverse_fun(params, inputs, **kw):
            for (_, _, ifun), p in zip(reversed(subs), reversed(list(params))):
                inputs, _ = ifun(p, inputs)
            return inputs, 0

        return params, direct_fun, inverse_fun


# --------------------------------------------------------------------------- distributions
class Normal(_InitFun):
    """flows/distributions.py:8-23 (prior spec only on the HIP path)"""

    def __init__(self, offset=0.0):
        self.offset = float(offset)
        self.spec = ("normal", self.offset)


class Uniform(_InitFun):
    """flows/distributions.py:26-41"""

    def __init__(self):
        self.spec = ("uniform",)


class Flow(_InitFun):
    """flows/distributions.py:67-112"""

    def __init__(self, transformation, prior=None, prior_support=None):
        self.transformation = transformation
        self.prior = prior if prior is not None else Normal()
        self.prior_support = prior_support
        if isinstance(self.prior, Uniform):
            if prior_support is None or tuple(float(v) for v in prior_support) != (0.0, 1.0):
                raise NotImplementedError("Uniform prior is built with prior_support=(0.0, 1.0) only (benchmark_tests.py:59-63)")
        elif isinstance(self.prior, Normal):
            if prior_support is not None:
                raise NotImplementedError("Normal prior with a support clip is not built")
        else:
            raise NotImplementedError("Flow priors: Normal(offset) or Uniform()")

    def __call__(self, rng, input_dim):
        params = self.transformation.init_params(rng, input_dim)
        if isinstance(self.prior, Uniform):
            kw = dict(prior=_lib.PRIOR_UNIFORM)
        else:
            kw = dict(prior=_lib.PRIOR_NORMAL, normal_offset=self.prior.offset)
        model = self.transformation.fused_model(input_dim, **kw)
        if model is None:
            raise NotImplementedError("Flow: the bijector stack must be [Box] + (IMADE|MADE, Reverse)*n")

        def log_pdf(params, inputs, return_sample=False):
            model.ensure_params(params)
            return model.log_pdf(inputs, return_sample=return_sample)

        def sample(rng, params, num_samples=1, return_original_samples=False):
            model.ensure_params(params)
            return model.sample(seed_from(rng), num_samples, return_latent=return_original_samples)

        log_pdf.model = sample.model = model
        return params, log_pdf, sample


class MFlow(_InitFun):
    """flows/distributions.py:116-194"""

    def __init__(self, transformation, sp_transformation, spline_degree, n_internal_knots, constraints_dict_left={0: 0},
                 constraints_dict_right={0: 0}, set_nn_output_grad_to_zero=False, n_spline_base_mesh_points=2000):
        self.gate = bool(set_nn_output_grad_to_zero)
        if sp_transformation.simple or sp_transformation.allow_negative_params:
            raise NotImplementedError("MFlow needs get_masked_transform() for the prior head")
        self.transformation, self.sp = transformation, sp_transformation
        self.k, self.n = spline_degree, n_internal_knots
        self.left, self.right = dict(constraints_dict_left), dict(constraints_dict_right)
        self.n_mesh = n_spline_base_mesh_points

    def __call__(self, rng, input_dim):
        g = as_generator(rng)
        tparams = self.transformation.init_params(g, input_dim)
        nb = self.n + self.k - 2  # msplines_jax.py:72-80: n_knots - k with k-fold end knots
        sparams = self.sp.init_params(g, input_dim, nb)
        model = self.transformation.fused_model(input_dim, prior=_lib.PRIOR_MFLOW, p_degree=self.k, p_knots=self.n,
                                                p_left=self.left, p_right=self.right, n_mesh=self.n_mesh, p_gate=self.gate)
        if model is None:
            raise NotImplementedError("MFlow: the bijector stack must be [Box] + (IMADE, Reverse)*n")
        assert model.p_nb == nb

        def log_pdf(params, inputs, return_sample=False):
            model.ensure_params(params)
            return model.log_pdf(inputs, return_sample=return_sample)

        def sample(rng, params, num_samples=1, return_original_samples=False, exact_inverse=False):
            model.ensure_params(params)
            return model.sample(seed_from(rng), num_samples, return_latent=return_original_samples, exact=exact_inverse)

        log_pdf.model = sample.model = model
        return (tparams, sparams), log_pdf, sample
