"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's EASE solver (rectorch/models.py:1015-1025 train,
:1054-1057 predict).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinned by tests/golden/g10_ease.npz (outputs of the reference's own EASE class run in the build container by
tests/golden/make_golden.py)."""
import numpy as np


def ease_fit(X, lam, timings=None):
    """B of models.py:1016-1024: G = X^T X; G[diag] += lam; P = inv(G); B = P / (-diag P); B[diag] = 0.
    ``timings`` (a dict) receives the wall time of the Gram product and of the inverse + scaling."""
    import time
    X = np.asarray(X, dtype=np.float64)
    t0 = time.perf_counter()
    G = X.T @ X
    t1 = time.perf_counter()
    idx = np.diag_indices(G.shape[0])
    G[idx] += lam
    P = np.linalg.inv(G)
    B = P / (-np.diag(P))
    B[idx] = 0
    if timings is not None:
        timings["gram_s"] = t1 - t0
        timings["inv_s"] = time.perf_counter() - t1
    return B


def ease_scores(X, B, ids, mask=None):
    """model[ids] with model = X B (models.py:1025, 1054) and -inf at mask.nonzero() (models.py:1055-1056)."""
    pred = np.asarray(X, dtype=np.float64)[ids] @ B
    if mask is not None:
        pred[np.asarray(mask).nonzero()] = -np.inf
    return pred
