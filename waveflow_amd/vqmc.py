"""waveflow.vqmc call surface on the HIP path (reference: vqmc.py:19-221): the VQMC trainer.

One step = sample a batch -> local energies -> gradient of loss_fn_efficient under its custom tangent rule -> Adam.
The device work (sampler, H psi, the vector-Jacobian product, the fp64 batch sums) is libwaveflow_hip; the optimiser
state lives on the host like the parameters do (wf_model_set_params takes the host vector), and with several ranks
the step's only collective is one SUM all-reduce of [gradient, sum E_L, sum E_L^2, n].

`adam` follows jax.example_libraries.optimizers.adam, which is what vqmc.py:136 builds:
    m <- (1 - b1) g + b1 m;  v <- (1 - b2) g^2 + b2 v;  x <- x - step_size * m_hat / (sqrt(v_hat) + eps)
with m_hat = m / (1 - b1^(i+1)), v_hat = v / (1 - b2^(i+1)) for the step index i passed to opt_update.
"""
import json
import os
from pathlib import Path

import numpy as np

from . import checkpoint
from .core import DeviceParams, flatten_params
from .model_factory import get_waveflow_model
from .utils import helpers, physics


class OptState:
    """Adam state over the flat parameter vector (reference leaf order); `template` restores the pytree.  x, m, v are numpy
    vectors (host state) or float32 cuda tensors (device state: parameters and optimiser never leave the GPU)."""

    def __init__(self, template, x, m, v, model=None):
        self.template, self.x, self.m, self.v, self.model = template, x, m, v, model
        self.version = 0

    @property
    def on_device(self):
        return self.model is not None


def adam(step_size, b1=0.9, b2=0.999, eps=1e-8, model=None):
    """-> (opt_init, opt_update, get_params), the triple protocol of jax.example_libraries.optimizers.
    With `model` (a DeviceModel) the state lives on that model's GPU, opt_update runs wf_adam_step in place and get_params
    returns core.DeviceParams; otherwise everything is numpy on the host."""
    def lr_at(i):
        return float(step_size(i) if callable(step_size) else step_size)

    def opt_init(params):
        x = flatten_params(params).astype(np.float32)
        if model is None:
            return OptState(params, x, np.zeros_like(x), np.zeros_like(x))
        import torch
        dev = f"cuda:{model.device}"
        template = params.template if isinstance(params, DeviceParams) else params
        xd = torch.as_tensor(x).to(dev)
        return OptState(template, xd, torch.zeros_like(xd), torch.zeros_like(xd), model=model)

    def opt_update(i, grads, state):
        if state.on_device:
            import torch
            g = grads if hasattr(grads, "is_cuda") else torch.as_tensor(flatten_params(grads) if not isinstance(grads, np.ndarray) else grads)
            g = g.to(state.x.device, dtype=torch.float32).contiguous()
            state.model.adam_step(state.x, g, state.m, state.v, i, lr_at(i), b1, b2, eps)
            state.version += 1
            return state
        g = grads if isinstance(grads, np.ndarray) and grads.ndim == 1 else flatten_params(grads)
        g = np.asarray(g, dtype=np.float32)
        one = np.float32(1.0)
        m = (one - np.float32(b1)) * g + np.float32(b1) * state.m
        v = (one - np.float32(b2)) * np.square(g) + np.float32(b2) * state.v
        mhat = m / (one - np.float32(b1) ** np.float32(i + 1))
        vhat = v / (one - np.float32(b2) ** np.float32(i + 1))
        x = state.x - np.float32(lr_at(i)) * mhat / (np.sqrt(vhat) + np.float32(eps))
        return OptState(state.template, x.astype(np.float32), m, v)

    def get_params(state):
        if state.on_device:
            return DeviceParams(state.template, state.x, state.version)
        return checkpoint.unflatten_like(state.template, state.x)

    return opt_init, opt_update, get_params


def create_train_state(box_length, learning_rate, n_particle, rng=0, xu_coord_type='mean', spline_degree=6, num_knots=23,
                       n_flow_layers=3):
    """vqmc.py:123-139"""
    init_fun = get_waveflow_model(n_particle, base_spline_degree=spline_degree, i_spline_degree=spline_degree,
                                  n_prior_internal_knots=num_knots, n_i_internal_knots=num_knots,
                                  i_spline_reg=0.05, i_spline_reverse_fun_tol=0.000001,
                                  n_flow_layers=n_flow_layers, box_size=box_length, xu_coord_type=xu_coord_type)
    params, psi, log_pdf, sample = init_fun(rng, n_particle)
    opt_init, opt_update, get_params = adam(step_size=learning_rate, model=psi.model)   # state on the model's GPU
    return psi, log_pdf, sample, opt_init(params), opt_update, get_params


def loss_fn_efficient(params, psi, h_fn, batch, running_average):
    """vqmc.py:193-200 (value only): mean of H psi / (psi + 1e-8) over the batch."""
    p = helpers._np(psi(params, batch)).reshape(-1, 1)
    e = helpers._np(h_fn(params, batch)).reshape(-1, 1)
    return float((e / (p + 1e-8)).mean())


def loss_and_grad_efficient(params, psi, h_fn, batch, running_average, group=None):
    """value_and_grad(loss_fn_efficient) with the custom tangent rule (vqmc.py:198-212, 215-221) over the walkers of ALL ranks
    of `group` (each rank passes its own shard).  -> (loss, flat gradient [n_params] float32 cuda tensor, (mean, variance,
    stderr) of E_L)."""
    from .distributed import all_reduce_gradient_and_moments, global_count, moments_to_stats
    model = psi.model
    model.ensure_params(params)
    pos = getattr(h_fn, "protons", None)
    if pos is None:
        raise TypeError("h_fn must come from waveflow_amd.utils.physics.construct_hamiltonian_function")
    # the tangent rule's 1 / batch factor is applied on the device, so the global count is needed up front
    # `group=None` means NOT distributed in every function of this module (as in ModelTrainer): no collective is entered, whatever
    # process group happens to be initialised -- a rank that passes None must not wait in an all-reduce its peers never reach.
    # Sharded walkers: pass the group explicitly (torch.distributed.group.WORLD for the default one).
    n_global = global_count(int(batch.shape[0]), f"cuda:{model.device}", group) if group is not None else int(batch.shape[0])
    sums, grad = model.vqmc_loss_grad(batch, pos, float(np.asarray(running_average).reshape(-1)[0]), global_count=n_global)
    if group is not None:
        grad, sums = all_reduce_gradient_and_moments(grad, sums, group)   # one collective per step
    s = sums.cpu().tolist()
    return s[0] / s[2], grad, moments_to_stats(s)


def train_step_efficient(epoch, psi, h_fn, opt_update, opt_state, params, batch, running_average, group=None):
    """vqmc.py:215-221 -> (new opt_state, loss)"""
    loss_val, gradients, _ = loss_and_grad_efficient(params, psi, h_fn, batch, running_average, group=group)
    return opt_update(epoch, gradients, opt_state), loss_val


def _energy_terms(params, psi, h_fn, batch):
    """One forward sweep: (model, x on the device, H psi, psi, laplacian) as float32 cuda vectors."""
    model = psi.model
    model.ensure_params(params)
    pos = getattr(h_fn, "protons", None)
    if pos is None:
        raise TypeError("h_fn must come from waveflow_amd.utils.physics.construct_hamiltonian_function")
    x, _ = model._to_dev(batch)
    hpsi, ps, lap = model.hamiltonian(x, pos, return_psi=True, return_laplacian=True)
    return model, x, hpsi, ps, lap


def _reduce_scalars(values, device, group):
    """SUM all-reduce of a few python floats (fp64 on the wire) -> list of floats; the identity for group=None (see loss_and_grad_efficient)."""
    if group is None:
        return [float(v) for v in values]
    import torch
    from .distributed import all_reduce_moments
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    return all_reduce_moments(t, group).cpu().tolist()


def _reduce_gradient(grad, group):
    if group is None:
        return grad
    import torch
    from .distributed import all_reduce_gradient_and_moments
    g, _ = all_reduce_gradient_and_moments(grad, torch.zeros(3, dtype=torch.float64, device=grad.device), group)
    return g


def loss_fn_uniform(params, psi, h_fn, batch):
    """vqmc.py:143-148: mean(psi * H psi) / mean(psi^2) over a uniformly drawn batch (value only; the gradient treats the
    denominator as a constant, see train_step_uniform)."""
    p = helpers._np(psi(params, batch)).reshape(-1, 1).astype(np.float64)
    e = helpers._np(h_fn(params, batch)).reshape(-1, 1).astype(np.float64)
    return float((p * e).mean() / (p * p).mean())


def loss_and_grad_uniform(params, psi, h_fn, batch, group=None):
    """value_and_grad(loss_fn_uniform) (vqmc.py:150-154) on the HIP path.  With H psi = -1/2 lap + V psi and the denominator
    Z = mean(psi^2) under stop_gradient,
        d loss = 1 / (n Z) * sum_b [ (H psi_b + V_b psi_b) d psi_b  -  1/2 psi_b d lap_b ],     V_b psi_b = H psi_b + 1/2 lap_b
    (no division by psi), which is one wf_psi_vjp.  Several ranks: n and Z are global (one small all-reduce), then the gradient."""
    model, x, hpsi, ps, lap = _energy_terms(params, psi, h_fn, batch)
    d = ps.double()
    num, den, n = _reduce_scalars([float((d * hpsi.double()).sum()), float((d * d).sum()), float(x.shape[0])], x.device, group)
    z = den / n
    scale = 1.0 / (n * z)
    w_psi = (2.0 * hpsi + 0.5 * lap) * scale
    w_lap = ps * (-0.5 * scale)
    grad = _reduce_gradient(model.psi_vjp(x, w_psi, w_lap), group)
    return (num / n) / z, grad


def train_step_uniform(epoch, psi, h_fn, opt_update, opt_state, get_params, batch, group=None):
    """vqmc.py:150-154 -> (new opt_state, loss)"""
    loss_val, gradients = loss_and_grad_uniform(get_params(opt_state), psi, h_fn, batch, group=group)
    return opt_update(epoch, gradients, opt_state), loss_val


def loss_fn(params, psi, h_fn, batch):
    """vqmc.py:157-161 -> (mean of H psi / psi, (H psi [B, 1], psi [B, 1]))"""
    p = helpers._np(psi(params, batch)).reshape(-1, 1)
    e = helpers._np(h_fn(params, batch)).reshape(-1, 1)
    return float((e / p).mean()), (e, p)


def train_step_gradients(params, psi, h_fn, log_pdf, batch, running_average, group=None, clip=10.0):
    """The gradient and loss of vqmc.train_step (vqmc.py:172-187) before the optimiser:
        energy part   grad of mean(H psi / psi) -- true derivative through psi and through the Laplacian:
                      d (E / psi) = -1/2 d lap / psi + (V / psi - E / psi^2) d psi = -1/(2 psi) d lap + lap / (2 psi^2) d psi
        density part  mean_b [ d log_pdf_b * (E_b / psi_b - running_average) ]   (the jacrev of the reference, contracted)
    summed, then clipped elementwise to [-clip, clip] (the reference's 10; None: unclipped, for tests); loss = mean of E / psi clipped to [-100, 100].
    -> (flat gradient float32 cuda [n_params], loss)"""
    import torch
    if getattr(log_pdf, "model", None) is not psi.model:
        raise ValueError("psi and log_pdf must be the two closures of one model (model_factory init_fun)")
    model, x, hpsi, ps, lap = _energy_terms(params, psi, h_fn, batch)
    ne = hpsi / ps
    loss_sum, n = _reduce_scalars([float(torch.clamp(ne, -100.0, 100.0).double().sum()), float(x.shape[0])], x.device, group)
    inv = 1.0 / n
    w_psi = 0.5 * lap / (ps * ps) * inv
    w_lap = -0.5 / ps * inv
    grad = model.psi_vjp(x, w_psi, w_lap)
    grad = grad + model.logpdf_vjp(x, (ne - float(np.asarray(running_average).reshape(-1)[0])) * inv)
    grad = _reduce_gradient(grad, group)
    return (torch.clamp(grad, -clip, clip) if clip is not None else grad), loss_sum / n


def train_step(epoch, psi, h_fn, log_pdf, opt_update, opt_state, get_params, batch, running_average, group=None):
    """vqmc.py:169-187 -> (new opt_state, loss)"""
    gradients, loss_val = train_step_gradients(get_params(opt_state), psi, h_fn, log_pdf, batch, running_average, group=group)
    return opt_update(epoch, gradients, opt_state), loss_val


def _save_optimizer_state(save_dir, opt_state, epoch):
    """Adam's moments next to the reference's artefacts (the reference checkpoints parameters only, vqmc.py:96-100, and a restart
    there begins with a fresh optimiser): `optimizer_state.npz` {m, v, epoch}, rewritten at every checkpoint."""
    m, v = (t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t) for t in (opt_state.m, opt_state.v))
    tmp = f"{save_dir}/optimizer_state.tmp.npz"
    np.savez(tmp, m=m, v=v, epoch=np.int64(epoch))
    os.replace(tmp, f"{save_dir}/optimizer_state.npz")


def _load_optimizer_state(save_dir, opt_state, epoch):
    """-> True when the moments of exactly this checkpoint epoch were restored."""
    path = f"{save_dir}/optimizer_state.npz"
    if not os.path.isfile(path):
        return False
    with np.load(path) as z:
        if int(z["epoch"]) != int(epoch) or z["m"].shape != tuple(opt_state.m.shape):
            return False
        m, v = z["m"].astype(np.float32), z["v"].astype(np.float32)
    if opt_state.on_device:
        import torch
        opt_state.m.copy_(torch.as_tensor(m))
        opt_state.v.copy_(torch.as_tensor(v))
    else:
        opt_state.m, opt_state.v = m, v
    return True


class ModelTrainer:
    """vqmc.py:19-119, same constructor arguments, attributes and on-disk artefacts."""

    def __init__(self, system_name='He', learning_rate=1e-4, box_length=10, num_epochs=200000, batch_size=128, log_every=2000):
        self.system_name = system_name
        self.n_space_dimension = 1
        self.system, self.n_particle = physics.system_catalogue[self.n_space_dimension][self.system_name]
        self.box_length = box_length
        self.xu_coord_type = 'mean'
        self.spline_degree = 6
        self.num_knots = 23
        self.n_flow_layer = 3
        self.realtime_plots = False
        self.n_plotting = 200
        self.log_every = log_every
        self.window = 100
        self.learning_rate = learning_rate
        self.num_epochs = num_epochs
        self.batch_size = batch_size
        self.save_dir = f'./results/{self.system_name}_{self.n_space_dimension}d_L{self.box_length}box'
        self.use_graph = True  # single process: capture the whole step (wf_vqmc_train_step) in a hipGraph and replay it
        self.seed = 2          # vqmc.py:57: PRNGKey(2)
        self.exact_sampler = False   # False: the reference's sampler (made.py:88 quirk); True: draws from |psi|^2

    def start_training(self, restart=False, verbose=True, group=None):
        """Under torch.distributed (one process per GPU, `torchrun examples/run_vqmc.py`) the walkers of a step are split over
        the ranks: every rank builds the same model from the same seed, draws batch_size / world walkers with its own sampler
        stream, and the step's single all-reduce (gradient + energy moments) keeps the replicas identical; rank 0 writes."""
        import torch.distributed as dist
        distributed = dist.is_available() and dist.is_initialized()
        if distributed and group is None:
            group = dist.group.WORLD   # (None would read as "not distributed" further down: the default group, explicitly)
        rank = dist.get_rank(group) if distributed else 0
        world = dist.get_world_size(group) if distributed else 1
        local_batch = (self.batch_size * (rank + 1)) // world - (self.batch_size * rank) // world
        verbose = verbose and rank == 0
        save_dir = self.save_dir
        rng = np.random.default_rng(self.seed)
        psi, log_pdf, sample, opt_state, opt_update, get_params = create_train_state(
            self.box_length, self.learning_rate, n_particle=self.n_particle, rng=int(rng.integers(1 << 31)),
            xu_coord_type=self.xu_coord_type, spline_degree=self.spline_degree, num_knots=self.num_knots, n_flow_layers=self.n_flow_layer)
        h_fn = physics.construct_hamiltonian_function(psi, protons=self.system, n_space_dimensions=self.n_space_dimension, eps=0.0)
        start_epoch = 0
        loss, energies = [0], []
        if Path(save_dir).is_dir() and restart:
            # unlike vqmc.py:68-71, which reloads the parameters but keeps stepping the freshly initialised optimiser state
            # (so the reloaded parameters are dropped after one step), a restart here continues from the checkpoint
            params, start_epoch = checkpoint.load_reference_checkpoint(f'{save_dir}/checkpoints')
            import torch
            opt_state.x.copy_(torch.as_tensor(flatten_params(params).astype(np.float32)))
            opt_state.version += 1
            if not _load_optimizer_state(save_dir, opt_state, start_epoch):
                # (a directory written by the reference, or by an older run): the moments restart from zero while the step index
                # -- and with it Adam's bias correction -- continues, so the first ~10 steps are up to 3x the nominal size
                import warnings
                warnings.warn(f"{save_dir}/optimizer_state.npz does not match checkpoint epoch {start_epoch}: Adam moments restart from zero")
            loss = np.load(f'{save_dir}/loss.npy').tolist()
            energies = np.load(f'{save_dir}/energies.npy').tolist()
        params = get_params(opt_state)
        running_average = np.zeros(1)
        system_dict = {"system_name": self.system_name, "box_length": self.box_length, "n_particle": self.n_particle,
                       "n_space_dimension": self.n_space_dimension, "window": self.window, "n_plotting": self.n_plotting}
        if rank == 0:
            helpers.make_result_dirs(save_dir)
            with open(f"{save_dir}/system_info.json", "w") as fout_sys:
                json.dump(system_dict, fout_sys, indent=4)
        if verbose:
            print("Start training...")
        # the whole step as one captured launch sequence, where the library has it (models the sweeps cover; the wave sampler: <= 131072 walkers
        # per step, no such limit where the staged large-batch sampler applies); otherwise the host-stepped loop below (sampler, loss + gradient, Adam: three library calls per step)
        # Sharded over several processes: the same sequence in two halves around the step's one all-reduce.
        fused = self.use_graph and local_batch >= 1
        if fused:
            from . import _lib
            fused = _lib.lib().wf_vqmc_train_step_workspace_bytes(psi.model._h, int(local_batch)) > 0
        if fused:
            params, loss, energies = self._train_graphed(psi, sample, h_fn, opt_state, get_params, start_epoch, loss, energies, system_dict,
                                                         save_dir, rng, verbose, group=group if distributed else None, rank=rank,
                                                         local_batch=local_batch)
            self.params, self.loss, self.energies = params, loss, energies
            self.psi, self.log_pdf, self.sample, self.h_fn = psi, log_pdf, sample, h_fn
            return params, loss
        for epoch in range(start_epoch + 1, start_epoch + self.num_epochs + 1):
            ckpt_seed, step_seed = int(rng.integers(1 << 31)), int(rng.integers(1 << 31))   # same host stream on every rank
            if (epoch % self.log_every == 0 or epoch == 1) and rank == 0:
                helpers.create_checkpoint_wavefunc(ckpt_seed, save_dir, psi, sample, params, epoch, loss, energies, system_dict)
                _save_optimizer_state(save_dir, opt_state, epoch)
            batch = sample(step_seed + 7919 * rank, params, local_batch, exact_inverse=self.exact_sampler)
            opt_state, new_loss = train_step_efficient(epoch, psi, h_fn, opt_update, opt_state, params, batch, running_average,
                                                       group=group)
            if epoch % 100 == 0:
                running_average = np.asarray(loss[-100:]).mean()
            if epoch % self.log_every == 0 and verbose:
                print(f"epoch {epoch} | Loss: {round(float(new_loss), 3)}")
            params = get_params(opt_state)
            loss.append(new_loss)
            energies.append([new_loss])
        self.params, self.loss, self.energies = params, loss, energies
        self.psi, self.log_pdf, self.sample, self.h_fn = psi, log_pdf, sample, h_fn
        return params, loss

    def _train_graphed(self, psi, sample, h_fn, opt_state, get_params, start_epoch, loss, energies, system_dict, save_dir, rng, verbose,
                       group=None, rank=0, local_batch=None):
        """The training loop with the step captured once in a hipGraph: per epoch one graph launch; the host looks at the
        losses every 100 epochs (to refresh the running average, vqmc.py:112-113) and at checkpoints.
        With `group` (one process per GPU) the step is wf_vqmc_train_step_local -> all-reduce of one packed fp64 buffer ->
        wf_vqmc_train_step_apply, issued call by call (WF_GRAPH_COLLECTIVE=1 captures it with the RCCL collective inside) -- either
        way without a host synchronisation per step."""
        import torch
        from . import _lib
        from .distributed import _all_reduce_sum
        model = psi.model
        # (the steps leave out the tables only the large-batch evaluation kernel reads; every evaluation below goes through
        # ensure_params / set_params_device first, and the loop ends with a full refresh)
        # Large batches are the exception: from WF_GRAD_TILE_MIN walkers per step on (default 16 384) the loss + gradient of the two-particle family runs
        # on the matrix cores (wf_kernels_etile.hip), which reads those tables -- the steps then refresh everything (tens of microseconds against
        # milliseconds of step)
        tile_min = int(os.environ.get("WF_GRAD_TILE_MIN", 16384))
        sample_min = int(os.environ.get("WF_SAMPLE_TILE_MIN", 16384))   # (the staged sampler of large batches reads them too)
        per_step = int(local_batch) if group is not None else int(self.batch_size)
        large = (tile_min > 0 and per_step >= tile_min) or (sample_min > 0 and per_step >= sample_min)
        st = model.make_train_state(opt_state.x, opt_state.m, opt_state.v, start_epoch + 1, ring_len=128, defer_eval_tables=not large)
        model.set_params_device(opt_state.x)
        seed = int(rng.integers(1 << 62))
        if group is None:
            def step():
                model.train_step(st, seed, self.batch_size, h_fn.protons, self.learning_rate, exact_sampler=self.exact_sampler)
            nbytes = _lib.check(_lib.lib().wf_vqmc_train_step_workspace_bytes(model._h, self.batch_size), "wf_vqmc_train_step_workspace_bytes")
            capture = True
        else:
            import torch.distributed as dist
            red = torch.zeros(model.n_params + 3, dtype=torch.float64, device=opt_state.x.device)

            def step():   # every rank draws its own walkers (stream seed + 7919 rank); the tangent rule carries 1 / global batch
                model.train_step_local(st, seed + 7919 * rank, local_batch, h_fn.protons, 1.0 / float(self.batch_size), red,
                                       exact_sampler=self.exact_sampler)
                _all_reduce_sum(red, group)
                model.train_step_apply(st, red, self.learning_rate)
            nbytes = _lib.check(_lib.lib().wf_vqmc_train_step_workspace_bytes(model._h, int(local_batch)), "wf_vqmc_train_step_workspace_bytes")
            # Capturing the collective in the graph measured no faster than issuing the three calls (4.91 vs 4.94 s per 20 000 steps, one
            # rank) and could not be tried on several GPUs here: opt-in.
            capture = dist.get_backend(group) == "nccl" and os.environ.get("WF_GRAPH_COLLECTIVE") == "1"
        st["ws"] = model._workspace(nbytes, opt_state.x.device)   # allocated here, outside the capture
        replay = step
        if capture:
            side = torch.cuda.Stream(device=model.device)
            side.wait_stream(torch.cuda.current_stream(model.device))
            try:
                with torch.cuda.stream(side):
                    if group is not None:   # RCCL sets up its communicator at the first collective: outside the capture
                        _all_reduce_sum(torch.zeros(8, dtype=torch.float64, device=opt_state.x.device), group)
                    side.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side):
                        step()
                replay = graph.replay
            except Exception as e:
                # a failed capture leaves the side stream in an undefined capture state: no silent fallback to call-by-call
                raise RuntimeError("hipGraph capture of the training step failed" + (" (WF_GRAPH_COLLECTIVE=1: the RCCL all-reduce inside "
                                   "the graph; unset it to issue the three calls per step instead)" if group is not None else "")) from e
            torch.cuda.current_stream(model.device).wait_stream(side)
        fetched = [start_epoch]   # losses of the epochs up to here are on the host

        def fetch(upto, keep_last_back=False):
            torch.cuda.synchronize(model.device)
            ring = st["ring"].cpu().numpy()
            new = [float(ring[e % st["ring_len"], 0] / ring[e % st["ring_len"], 2]) for e in range(fetched[0] + 1, upto + 1)]
            fetched[0] = upto
            return new

        params = get_params(opt_state)
        for epoch in range(start_epoch + 1, start_epoch + self.num_epochs + 1):
            if epoch % self.log_every == 0 or epoch == 1:
                new = fetch(epoch - 1)
                loss.extend(new)
                energies.extend([[v] for v in new])
                opt_state.version += 1
                params = get_params(opt_state)
                ckpt_seed = int(rng.integers(1 << 31))   # (drawn on every rank: the host streams stay aligned)
                if rank == 0:
                    helpers.create_checkpoint_wavefunc(ckpt_seed, save_dir, psi, sample, params, epoch, loss, energies, system_dict)
                    _save_optimizer_state(save_dir, opt_state, epoch)   # (fetch() above synchronised: m, v are those after epoch - 1)
                model.set_params_device(opt_state.x)
            replay()
            if epoch % 100 == 0:
                new = fetch(epoch)
                loss.extend(new[:-1])
                energies.extend([[v] for v in new[:-1]])
                st["running_average"].fill_(float(np.asarray(loss[-100:]).mean()))   # before this epoch's loss joins (vqmc.py:112-118)
                loss.append(new[-1])
                energies.append([new[-1]])
                if epoch % self.log_every == 0 and verbose:
                    print(f"epoch {epoch} | Loss: {round(float(new[-1]), 3)}")
        new = fetch(start_epoch + self.num_epochs)
        loss.extend(new)
        energies.extend([[v] for v in new])
        opt_state.version += 1
        model.set_params_device(opt_state.x)   # every image of the model, evaluation tables included, holds the final parameters
        return get_params(opt_state), loss, energies
