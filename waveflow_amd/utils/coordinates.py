"""Host-side coordinate helpers (reference: utils/coordinates.py:41-51)."""
import numpy as np


def get_num_inversion_count(coordinates):
    """Per row, the number of inversions needed to sort it (sign of the antisymmetrised psi is (-1)**count)."""
    c = np.asarray(coordinates)
    B, D = c.shape
    count = np.zeros(B, dtype=np.int64)
    for i in range(D):
        for j in range(i + 1, D):
            count += c[:, i] > c[:, j]
    return count
