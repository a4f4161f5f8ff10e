"""Coordinate helpers (reference: utils/coordinates.py:41-51)."""
import numpy as np


def get_num_inversion_count(coordinates):
    """Per row, the number of inversions needed to sort it (sign of the antisymmetrised psi is (-1)**count).
    A cuda tensor is counted on the device (wf_inversion_count) and returned as an int32 cuda tensor; anything else on the host,
    one vectorised comparison of all pairs (the reference's per-row insertion loop, coordinates.py:17-51, counts the same pairs)."""
    if hasattr(coordinates, "is_cuda") and coordinates.is_cuda:
        import ctypes

        import torch

        from .. import _lib
        x = coordinates.to(torch.float32).contiguous()
        B, D = x.shape
        out = torch.empty(B, dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().wf_inversion_count(ctypes.c_void_p(x.data_ptr()) if B else None, B, D, ctypes.c_void_p(out.data_ptr()) if B else None,
                                                     ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "wf_inversion_count")
        return out
    c = np.asarray(coordinates)
    i, j = np.triu_indices(c.shape[1], k=1)
    return (c[:, i] > c[:, j]).sum(axis=1).astype(np.int64)
