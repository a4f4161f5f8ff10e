"""waveflow.utils.physics call surface on the HIP path (reference: utils/physics.py)."""
import numpy as np

# physics.system_catalogue (physics.py:6-26): system name -> (proton positions, number of electrons), by space dimension
system_catalogue = {
    1: {
        'Laplacian_interactive_particles': (np.array([]), 2),
        'H': (np.array([[0.0]]), 1),
        'He+': (np.array([[0.0], [0.0]]), 1),
        'H2+': (np.array([[-0.9], [0.9]]), 1),
        'H2+_wide': (np.array([[-3.0], [3.0]]), 1),
        'He': (np.array([[0.0], [0.0]]), 2),
        'He_off_center': (np.array([[2.5], [2.5]]), 2),
        'H2': (np.array([[-0.9], [0.9]]), 2),
        'H2_wide': (np.array([[-3.0], [3.0]]), 2),
    },
}


def construct_hamiltonian_function(fn, protons=np.array([[0, 0]]), n_space_dimensions=1, eps=0.0, max_potential_val=None):
    """physics.py:79-93.  `fn` must be the psi closure of a waveflow_amd Waveflow model; returns h_fn(params, x) -> [B, 1]
    with H psi = -0.5 * laplacian(psi) + V(x) * psi (autodiff Laplacian, eps = 0)."""
    if n_space_dimensions != 1:
        raise NotImplementedError("the reference's potential is one-dimensional only (physics.py:62 TODO)")
    if eps != 0.0:
        raise NotImplementedError("numerical Laplacian (eps > 0) is not built; eps = 0 is what vqmc.py:65-66 uses")
    model = getattr(fn, "model", None)
    if model is None:
        raise TypeError("fn must be the psi closure returned by waveflow_amd.wavefunctions.Waveflow")
    pos = np.asarray(protons, dtype=np.float32).reshape(-1)

    def h_fn(params, x):
        model.ensure_params(params)
        h = model.hamiltonian(x, pos)
        return h[:, None]

    h_fn.model = model
    h_fn.protons = pos
    return h_fn
