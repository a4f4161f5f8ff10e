"""Independent check of the training target: lowest antisymmetric eigenvalue of the 1-D soft-Coulomb helium Hamiltonian
(physics.py:60-93: -1/2 sum d^2/dx_i^2 - 2 sum 1/sqrt(1+x_i^2) + 1/sqrt(1+(x1-x2)^2)) in the box [-L, L]^2, finite differences."""
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla
for L, n in ((10.0, 300), (10.0, 400), (12.0, 400)):
    x = np.linspace(-L, L, n + 2)[1:-1]          # Dirichlet walls
    h = x[1] - x[0]
    T1 = sp.diags([-0.5 / h**2, 1.0 / h**2, -0.5 / h**2], [-1, 0, 1], shape=(n, n))   # -1/2 d^2/dx^2
    I = sp.identity(n)
    X1, X2 = np.meshgrid(x, x, indexing="ij")
    V = -2 / np.sqrt(1 + X1**2) - 2 / np.sqrt(1 + X2**2) + 1 / np.sqrt(1 + (X1 - X2)**2)
    H = sp.kron(T1, I) + sp.kron(I, T1) + sp.diags(V.ravel())
    w, v = sla.eigsh(H.tocsc(), k=4, which="SA")
    out = []
    for e, vec in zip(w, v.T):
        P = vec.reshape(n, n)
        sym = np.linalg.norm(P - P.T) / np.linalg.norm(P)     # ~0: symmetric (singlet), ~2: antisymmetric (triplet)
        out.append((round(float(e), 5), "antisym" if sym > 1 else "sym"))
    print(f"L={L} n={n} h={h:.4f}:", out)
