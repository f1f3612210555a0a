r"""Batch samplers with the reference's API (rectorch/samplers.py:18-107) and a device-resident fast path.

``DataSampler`` keeps the rating matrices in HBM as CSR (uploaded once) and forms every batch on the
MI355X with the LDS-staged gather kernel, instead of the reference's per-batch scipy fancy-index +
``toarray()`` + float64->float32 copy on the host (the reference's CPU hot spot #1, ~164 ms per 500-user
batch).  Iterating it yields the same ``(data_tr, data_te | None)`` pairs of float32 ``[B, n_items]``
tensors in the same order (the permutation comes from the same global-numpy-RNG shuffle), now resident on the
device; ``iter_rows()`` yields only row numbers for trainers that take the sparse rows directly.
"""
import numpy as np
import torch
from scipy.sparse import csr_matrix, hstack

from .engine import CsrMatrix, RowBatch, SvaePack, SvaeTarget, tag_rows

__all__ = ['Sampler', 'DataSampler', 'ConditionedDataSampler', 'BalancedConditionedDataSampler',
           'EmptyConditionedDataSampler', 'SVAE_Sampler']


class Sampler():
    r"""Sampler base class (reference samplers.py:18-40): a generator of batches.  Sub-classes implement
    ``__len__`` (number of batches) and ``__iter__``."""
    def __init__(self, *args, **kargs):
        pass

    def __len__(self):
        """Return the number of batches."""
        raise NotImplementedError

    def __iter__(self):
        """Iterate through the batches yielding a batch at a time."""
        raise NotImplementedError


class DataSampler(Sampler):
    r"""Plain batches of users, in order or shuffled (counterpart of reference samplers.py:43-107).

    Arguments (the first four are the reference's, positionally compatible; the attributes of the same names are public):

    * ``sparse_data_tr`` -- scipy CSR matrix, one row per user: what the model reads.
    * ``sparse_data_te`` -- optional CSR matrix of the same users: the second element of every yielded pair (held-out
      items at evaluation time); ``None`` yields ``None`` there.
    * ``batch_size`` -- users per batch (the last batch may be shorter); default 1.
    * ``shuffle`` -- draw a fresh permutation of the users from numpy's GLOBAL generator at every ``iter()``; default on.
    * ``device`` (not in the reference) -- ``None`` picks the MI355X when one is visible: the matrices are uploaded once
      and batches are formed by the gather kernel; ``"cpu"`` reproduces the reference's host behaviour (scipy slicing +
      ``toarray()``) for callers without a HIP device.  It is not a compute path of this package.
    """
    def __init__(self,
                 sparse_data_tr,
                 sparse_data_te=None,
                 batch_size=1,
                 shuffle=True,
                 device=None):
        super(DataSampler, self).__init__()
        self.sparse_data_tr = sparse_data_tr
        self.sparse_data_te = sparse_data_te
        self.batch_size = batch_size
        self.shuffle = shuffle
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        self._csr_tr = self._csr_te = None
        self._src = (None, None)

    def __len__(self):
        return int(np.ceil(self.sparse_data_tr.shape[0] / self.batch_size))

    @property
    def resident(self):
        return self.device.type == "cuda"

    def _upload(self):
        # re-upload if the user swapped the matrices (attributes are public, as in the reference)
        if self._src[0] is not self.sparse_data_tr:
            self._csr_tr = CsrMatrix(self.sparse_data_tr)
        if self._src[1] is not self.sparse_data_te:
            self._csr_te = None if self.sparse_data_te is None else CsrMatrix(self.sparse_data_te)
            if self._csr_te is not None:
                # same users; a conditioned input matrix carries its condition columns after the items (CMultiVAE)
                assert self._csr_te.shape[0] == self._csr_tr.shape[0] and self._csr_te.shape[1] <= self._csr_tr.shape[1], \
                    "tr and te matrices must have the same shape"
        self._src = (self.sparse_data_tr, self.sparse_data_te)

    def _order(self):
        n = self.sparse_data_tr.shape[0]
        idxlist = list(range(n))
        if self.shuffle:
            np.random.shuffle(idxlist)      # global numpy RNG, fresh permutation per iter() (samplers.py:93-95)
        return n, idxlist

    def iter_rows(self):
        """Fast path: yields :class:`rectorch_amd.engine.RowBatch` (row numbers on the device), nothing dense."""
        assert self.resident, "iter_rows() needs the device-resident sampler"
        self._upload()
        if self.shuffle:
            n, idxlist = self._order()
            rows = torch.from_numpy(np.asarray(idxlist, dtype=np.int32)).to(self.device)
        else:
            # the identity order of a validation / test loader: one device arange, kept (a Python list of n ints turned into an array
            # and copied to the device was 250 us of every evaluate() call -- a tenth of the whole call at 10 000 users)
            n = self.sparse_data_tr.shape[0]
            rows = getattr(self, "_seq_rows", None)
            if rows is None or rows.numel() != n or rows.device != self.device:
                rows = self._seq_rows = torch.arange(n, dtype=torch.int32, device=self.device)
        for start_idx in range(0, n, self.batch_size):
            end_idx = min(start_idx + self.batch_size, n)
            yield RowBatch(self._csr_tr, self._csr_te, rows[start_idx:end_idx])

    def __iter__(self):
        if not self.resident:
            yield from self._iter_host()
            return
        for rb in self.iter_rows():
            # tagged: the trainer skips the dense detour when it gets this tensor back UNMODIFIED
            data_tr = tag_rows(self._csr_tr.gather_dense(rb.rows), rb)
            data_te = None
            if self._csr_te is not None:
                data_te = tag_rows(self._csr_te.gather_dense(rb.rows), rb)
            yield data_tr, data_te

    def _iter_host(self):
        # the reference's host behaviour, for callers without a HIP device (samplers.py:91-107)
        n, idxlist = self._order()
        for start_idx in range(0, n, self.batch_size):
            end_idx = min(start_idx + self.batch_size, n)
            data_tr = self.sparse_data_tr[idxlist[start_idx:end_idx]]
            data_tr = torch.FloatTensor(data_tr.toarray())
            data_te = None
            if self.sparse_data_te is not None:
                data_te = self.sparse_data_te[idxlist[start_idx:end_idx]]
                data_te = torch.FloatTensor(data_te.toarray())
            yield data_tr, data_te


def _sparse_pair(data_tr, data_te):
    """``(input rows, target rows)`` of one batch as two device-resident :class:`RowBatch` objects over the same row
    ids -- what the ``sparse=True`` samplers yield in place of the dense ``(data, target)`` tensors.  It is a PAIR like
    every other sampler's item: ``train_epoch`` / ``evaluate`` / ``one_plus_random`` unpack it unchanged."""
    tr, te = CsrMatrix(data_tr), CsrMatrix(data_te)
    rows = torch.arange(data_tr.shape[0], dtype=torch.int32, device="cuda")
    return RowBatch(tr, te, rows), RowBatch(te, None, rows)


class ConditionedDataSampler(Sampler):
    r"""Data sampler with conditioned filtering for :class:`rectorch_amd.models.CMultiVAE` (reference
    samplers.py:108-234).

    Every user appears once unconditioned -- example ``(row, -1)`` -- and once per condition known to at least one of
    the user's items -- ``(row, c)``.  A batch is the examples' training rows with the ``n_cond`` one-hot condition
    columns appended, and as target the test rows restricted to the items valid under the example's condition (all
    items having any condition when unconditioned); examples whose filtered target is empty are dropped.

    This is host-side bookkeeping, as in the reference; the batches it yields are what the MI355X engine consumes.
    With ``sparse=True`` (not in the reference) the pair is yielded as two :class:`rectorch_amd.engine.RowBatch`
    objects (input rows, target rows) over small per-batch CSR matrices uploaded to HBM, so nothing dense of width
    ``n_items`` crosses PCIe.

    Arguments: ``iid2cids`` (dict item id -> list of the conditions, integers below ``n_cond``, the item satisfies),
    ``n_cond`` (how many conditions exist), ``sparse_data_tr`` / ``sparse_data_te`` (CSR matrices of the users' input and
    target items; the target defaults to the input), ``batch_size`` (examples per batch, default 1), ``shuffle``
    (permute the examples with numpy's global generator before batching, default on), ``sparse`` (see above).
    """
    def __init__(self,
                 iid2cids,
                 n_cond,
                 sparse_data_tr,
                 sparse_data_te=None,
                 batch_size=1,
                 shuffle=True,
                 sparse=False):
        super(ConditionedDataSampler, self).__init__()
        self.sparse_data_tr = sparse_data_tr
        self.sparse_data_te = sparse_data_te
        self.iid2cids = iid2cids
        self.batch_size = batch_size
        self.n_cond = n_cond
        self.shuffle = shuffle
        self.sparse = sparse
        self._compute_conditions()

    def _row_conditions(self):
        """row -> the set of conditions of the row's items.  Built as a union of per-item sets in column order so that
        iterating it visits the conditions in the order the reference's own sets do (samplers.py:171-174)."""
        tr = self.sparse_data_tr.tocsr()
        out = {}
        for r in range(tr.shape[0]):
            cols = tr.indices[tr.indptr[r]:tr.indptr[r + 1]][tr.data[tr.indptr[r]:tr.indptr[r + 1]] != 0]
            out[r] = set.union(*[set(self.iid2cids[c]) for c in np.sort(cols)])
        return out

    def _item_condition_matrix(self):
        items = list(self.iid2cids)
        rows = np.repeat(items, [len(self.iid2cids[m]) for m in items]).astype(np.int64)
        cols = np.array([g for m in items for g in self.iid2cids[m]], dtype=np.int64)
        return csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(len(self.iid2cids), self.n_cond))

    def _compute_conditions(self):
        r2cond = self._row_conditions()
        ex = [(r, -1) for r in r2cond]
        ex += [(r, c) for r in r2cond for c in r2cond[r]]
        self.examples = np.array(ex)
        self.M = self._item_condition_matrix()

    def __len__(self):
        return int(np.ceil(len(self.examples) / self.batch_size))

    def _batch(self, ex):
        """(input rows [b, n_items + n_cond], filtered target rows [b, n_items]) as scipy CSR, empty targets dropped"""
        n = len(ex)
        rows_, conds = ex[:, 0], ex[:, 1]
        has = conds >= 0
        onehot = csr_matrix((np.ones(int(has.sum())), (np.nonzero(has)[0], conds[has])), shape=(n, self.n_cond))
        data_tr = hstack([self.sparse_data_tr[rows_], onehot], format="csr")
        if self.sparse_data_te is None:
            self.sparse_data_te = self.sparse_data_tr
        # an unconditioned example admits every condition
        allow = onehot.tolil()
        allow[np.nonzero(~has)[0], :] = 1
        filtered = allow.tocsr().dot(self.M.transpose().tocsr()) > 0
        data_te = self.sparse_data_te[rows_].multiply(filtered).tocsr()
        keep = np.diff(data_te.indptr) != 0
        return data_tr[keep], data_te[keep]

    def _emit(self, data_tr, data_te):
        if self.sparse:
            return _sparse_pair(data_tr, data_te)
        return torch.FloatTensor(data_tr.toarray()), torch.FloatTensor(data_te.toarray())

    def __iter__(self):
        n = len(self.examples)
        idxlist = list(range(n))
        if self.shuffle:
            np.random.shuffle(idxlist)
        for start_idx in range(0, n, self.batch_size):
            end_idx = min(start_idx + self.batch_size, n)
            data_tr, data_te = self._batch(self.examples[idxlist[start_idx:end_idx]])
            yield self._emit(data_tr, data_te)


class BalancedConditionedDataSampler(ConditionedDataSampler):
    r"""Sub-sampled version of :class:`ConditionedDataSampler` (reference samplers.py:237-338): every user once
    unconditioned, plus for each condition *c* ``m = int(num_cond_examples * subsample / n_cond)`` users drawn with
    replacement (``np.random.choice``) among those knowing *c*.

    ``subsample`` (in (0, 1], default 0.2) is the fraction of conditioned examples kept; the other arguments are
    :class:`ConditionedDataSampler`'s, without ``shuffle`` (the reference has none here either).
    """
    def __init__(self,
                 iid2cids,
                 n_cond,
                 sparse_data_tr,
                 sparse_data_te=None,
                 batch_size=1,
                 subsample=.2,
                 sparse=False):
        super(BalancedConditionedDataSampler, self).__init__(iid2cids,
                                                             n_cond,
                                                             sparse_data_tr,
                                                             sparse_data_te,
                                                             batch_size,
                                                             sparse=sparse)
        self.subsample = subsample
        self._compute_sampled_conditions()

    def _compute_conditions(self):
        r2cond = self._row_conditions()
        self.examples = {-1: list(r2cond.keys())}
        for c in range(self.n_cond):
            self.examples[c] = [r for r in r2cond if c in r2cond[r]]
        self.num_cond_examples = sum(len(self.examples[c]) for c in range(self.n_cond))
        self.M = self._item_condition_matrix()

    def _compute_sampled_conditions(self):
        data = [(r, -1) for r in self.examples[-1]]
        m = int(self.num_cond_examples * self.subsample / self.n_cond)
        for c in range(self.n_cond):
            data += [(r, c) for r in np.random.choice(self.examples[c], m)]
        self.examples = np.array(data)

    def __len__(self):
        m = int(self.num_cond_examples * self.subsample) + self.sparse_data_tr.shape[0]
        return int(np.ceil(m / self.batch_size))


class EmptyConditionedDataSampler(Sampler):
    r"""Unconditioned batches for :class:`rectorch_amd.models.CMultiVAE` (reference samplers.py:341-419): like
    :class:`DataSampler`, with ``cond_size`` zero columns appended to the input rows.

    ``cond_size`` is the number of condition columns to append; the remaining arguments are :class:`DataSampler`'s.
    """
    def __init__(self,
                 cond_size,
                 sparse_data_tr,
                 sparse_data_te=None,
                 batch_size=1,
                 shuffle=True,
                 sparse=False):
        super(EmptyConditionedDataSampler, self).__init__()
        self.sparse_data_tr = sparse_data_tr
        self.sparse_data_te = sparse_data_te
        self.batch_size = batch_size
        self.cond_size = cond_size
        self.shuffle = shuffle
        self.sparse = sparse

    def __len__(self):
        return int(np.ceil(self.sparse_data_tr.shape[0] / self.batch_size))

    def __iter__(self):
        n = self.sparse_data_tr.shape[0]
        idxlist = list(range(n))
        if self.shuffle:
            np.random.shuffle(idxlist)
        for start_idx in range(0, n, self.batch_size):
            rows = idxlist[start_idx:min(start_idx + self.batch_size, n)]
            data_tr = self.sparse_data_tr[rows]
            data_tr = hstack([data_tr, csr_matrix((data_tr.shape[0], self.cond_size))], format="csr")
            if self.sparse_data_te is None:
                self.sparse_data_te = self.sparse_data_tr
            data_te = self.sparse_data_te[rows]
            if self.sparse:
                yield _sparse_pair(data_tr, data_te)
            else:
                yield torch.FloatTensor(data_tr.toarray()), torch.FloatTensor(data_te.toarray())


class SVAE_Sampler(Sampler):
    r"""Sampler of user sequences for :class:`rectorch_amd.models.SVAE` (reference samplers.py:446-571): one user per
    batch, ``x`` = the LongTensor ``[1, T]`` of all but the user's last item, ``y`` = the multi-hot targets
    ``[1, T, num_items]``: at step *t* the next item (``'next'``), the next ``k`` items (``'next_k'``) or all the
    remaining ones (``'postfix'``); with ``is_training=False`` a single row ``[1, 1, num_items]`` holding the user's
    test items.

    With ``sparse=True`` (not in the reference) ``y`` is a :class:`rectorch_amd.engine.SvaeTarget` -- the same rows as
    a CSR on the device -- so no ``T x num_items`` dense tensor is built or copied per user.

    Arguments: ``num_items``; ``dict_data_tr`` (dict user -> list of item ids in interaction order, users numbered from
    0); ``dict_data_te`` (dict user -> test items; required when ``is_training`` is off); ``pred_type`` (``'next_k'``
    -- the default -- ``'next'`` or ``'postfix'``); ``k`` (items to predict for ``'next_k'``, at least 1); ``shuffle``
    (visit the users in a random order drawn from numpy's global generator, default on); ``is_training`` (default on);
    ``sparse`` (see above); ``pack`` (not in the reference, default 1 = one user per batch and per optimizer step, as the
    reference): with ``pack = N > 1`` a training batch is a :class:`rectorch_amd.engine.SvaePack` of up to ``N`` users --
    yielded as ``(pack, pack)`` -- on which :meth:`rectorch_amd.models.SVAE.train_batch` takes ONE Adam step for the mean of the
    users' losses (gradient accumulation over the pack).  A pack costs its LONGEST recurrence, so the (shuffled) users are
    grouped by length inside windows of ``16 * N`` users and the packs of a window visited in random order; ``pack_tokens``
    bounds the time steps of one pack (default 16 384).
    """
    def __init__(self,
                 num_items,
                 dict_data_tr,
                 dict_data_te=None,
                 pred_type="next_k",
                 k=1,
                 shuffle=True,
                 is_training=True,
                 sparse=False,
                 pack=1,
                 pack_tokens=16384):
        super(SVAE_Sampler, self).__init__()
        if pred_type == "next_k":
            assert k >= 1, "If pred_type == 'next_k' then 'k' must be a positive integer."
        self.pred_type = pred_type
        self.dict_data_tr = dict_data_tr
        self.dict_data_te = dict_data_te
        self.shuffle = shuffle
        self.num_items = num_items
        self.k = k
        self.is_training = is_training
        self.sparse = sparse
        assert pack >= 1 and pack_tokens >= 1
        self.pack = int(pack)
        self.pack_tokens = int(pack_tokens)

    def __len__(self):
        if self.pack > 1 and self.is_training:
            # packs of the unshuffled order; a shuffled epoch may differ by a pack or two (the token bound cuts differently)
            return sum(len(w) for w in self._pack_windows(list(range(len(self.dict_data_tr)))))
        return len(self.dict_data_tr)

    def _pack_windows(self, idxlist):
        """the users of ``idxlist`` cut into windows of 16 * pack, each sorted by length and cut into packs (lists of users)
        of at most ``pack`` users and ``pack_tokens`` time steps; users without a time step (fewer than 2 items) are skipped"""
        window = 16 * self.pack
        out = []
        for w0 in range(0, len(idxlist), window):
            win = [u for u in idxlist[w0:w0 + window] if len(self.dict_data_tr[u]) >= 2]
            win.sort(key=lambda u: len(self.dict_data_tr[u]))
            packs, cur, tok = [], [], 0
            for u in win:
                t = len(self.dict_data_tr[u]) - 1
                if cur and (len(cur) == self.pack or tok + t > self.pack_tokens):
                    packs.append(cur)
                    cur, tok = [], 0
                cur.append(u)
                tok += t
            if cur:
                packs.append(cur)
            out.append(packs)
        return out

    def _target_rows(self, user):
        """the distinct target items of every time step of ``user`` (list of lists)"""
        seq = self.dict_data_tr[user]
        if not self.is_training:
            return [list(dict.fromkeys(self.dict_data_te[user]))]
        rows = []
        for t in range(len(seq) - 1):
            if self.pred_type == 'next':
                r = [seq[t + 1]]
            elif self.pred_type == 'next_k':
                r = seq[t + 1:][:self.k]
            elif self.pred_type == 'postfix':
                r = seq[t + 1:]
            else:
                r = []
            rows.append(list(dict.fromkeys(r)))
        return rows

    def __iter__(self):
        idxlist = list(range(len(self.dict_data_tr)))
        if self.shuffle:
            np.random.shuffle(idxlist)
        if self.pack > 1 and self.is_training:
            for packs in self._pack_windows(idxlist):
                order = np.random.permutation(len(packs)) if self.shuffle else range(len(packs))
                for pi in order:
                    users = packs[pi]
                    p = SvaePack([self.dict_data_tr[u][:-1] for u in users], [self._target_rows(u) for u in users], users)
                    yield p, p
            return
        for user in idxlist:
            rows = self._target_rows(user)
            x = torch.LongTensor([self.dict_data_tr[user][:-1]])
            if self.sparse:
                yield x, SvaeTarget(rows)
                continue
            y = torch.zeros(1, len(rows), self.num_items)
            for t, r in enumerate(rows):
                y[0, t, r] = 1.
            yield x, y
