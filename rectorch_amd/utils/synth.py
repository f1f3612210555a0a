"""Synthetic implicit-feedback matrices shaped like the reference's datasets.

The reference ships no data (datasets are git-ignored) and the GPU box has no network, so benchmarks
and full-size parity tests use a seeded generator: per-user degree ~ clipped log-normal, item ids
drawn from a Zipf-like popularity law, binary values, CSR float64 like
``DataReader._load_train_data`` builds (reference data.py:363-377).
"""
import numpy as np
from scipy.sparse import csr_matrix


def synth_interactions(n_users, n_items, mu=3.9, sigma=0.9, dmin=5, dmax=2000, zipf_a=0.9,
                       seed=20240927, dtype=np.float64):
    """Return a ``scipy.sparse.csr_matrix`` [n_users, n_items] with sorted, unique column ids.

    ml-20m shape: ``synth_interactions(116677, 20108)`` (mean degree ~74);
    Netflix shape: ``synth_interactions(480000, 17769, mu=4.3, sigma=1.0, dmax=5000)``.
    """
    rng = np.random.default_rng(seed)
    dmax = min(dmax, n_items)
    deg = np.clip(np.rint(rng.lognormal(mu, sigma, n_users)), min(dmin, n_items), dmax).astype(np.int64)
    p = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), zipf_a)
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    # oversample with replacement, dedupe per user, trim to the target degree
    over = (deg * 1.6).astype(np.int64) + 16
    starts = np.concatenate([[0], np.cumsum(over)])
    draws = np.searchsorted(cdf, rng.random(int(starts[-1])), side="right").astype(np.int64)
    draws = np.minimum(draws, n_items - 1)
    owner = np.repeat(np.arange(n_users, dtype=np.int64), over)
    key = owner * n_items + draws
    # dedupe keeping the FIRST occurrence in draw order (draws are laid out user by user), keep the
    # first deg[u] distinct items of each user, then sort each user's items by id
    _, first_pos = np.unique(key, return_index=True)
    first_pos.sort()
    key = key[first_pos]
    owner = key // n_items
    cnt = np.bincount(owner, minlength=n_users)
    first = np.concatenate([[0], np.cumsum(cnt)])[:-1]
    rank_in_user = np.arange(key.size) - np.repeat(first, cnt)
    key = np.sort(key[rank_in_user < np.repeat(deg, cnt)])
    owner = key // n_items
    item = key % n_items
    cnt = np.bincount(owner, minlength=n_users)
    indptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    data = np.ones(item.size, dtype=dtype)
    return csr_matrix((data, item.astype(np.int32), indptr), shape=(n_users, n_items))


def split_heldout(csr, test_prop=0.2, seed=98765):
    """Per-user 80/20 item split of a held-out user block (reference data.py:251-272 rule:
    users with >=5 items give ``test_prop`` of them to the te part)."""
    rng = np.random.default_rng(seed)
    indptr, indices = csr.indptr, csr.indices
    n = csr.shape[0]
    tr_rows, tr_cols, te_rows, te_cols = [], [], [], []
    for u in range(n):
        cols = indices[indptr[u]:indptr[u + 1]]
        if cols.size >= 5:
            mask = np.zeros(cols.size, dtype=bool)
            k = int(test_prop * cols.size)
            mask[rng.choice(cols.size, size=k, replace=False)] = True
            te_rows.append(np.full(int(mask.sum()), u)); te_cols.append(cols[mask])
            tr_rows.append(np.full(int((~mask).sum()), u)); tr_cols.append(cols[~mask])
        else:
            tr_rows.append(np.full(cols.size, u)); tr_cols.append(cols)
    def build(rows, cols):
        r = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        c = np.concatenate(cols) if cols else np.zeros(0, np.int64)
        return csr_matrix((np.ones(r.size), (r, c)), shape=csr.shape)
    return build(tr_rows, tr_cols), build(te_rows, te_cols)
