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

from .engine import CsrMatrix, RowBatch

__all__ = ['Sampler', 'DataSampler']


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
    r"""Standard sampler returning batches without any particular constraint (reference samplers.py:43-107).

    Parameters
    ----------
    sparse_data_tr : :obj:`scipy.sparse.csr_matrix`
        The training sparse user-item rating matrix.
    sparse_data_te : :obj:`scipy.sparse.csr_matrix` [optional]
        The test sparse user-item rating matrix (same shape), by default :obj:`None`.
    batch_size : :obj:`int` [optional]
        The size of the batches, by default 1.
    shuffle : :obj:`bool` [optional]
        Whether the data set must be randomly shuffled before creating the batches, by default ``True``.
    device : :obj:`str` / :class:`torch.device` / ``None`` [optional, not in the reference]
        ``None``: the MI355X if one is visible (resident CSR + gather kernel), else host tensors.
        ``"cpu"`` forces the reference's host behaviour (yields ``torch.FloatTensor`` built with scipy);
        it exists for host-only callers of the API and is not a compute path of this package.
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
                assert self._csr_te.shape == self._csr_tr.shape, "tr and te matrices must have the same shape"
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
        n, idxlist = self._order()
        rows = torch.from_numpy(np.asarray(idxlist, dtype=np.int32)).to(self.device)
        for start_idx in range(0, n, self.batch_size):
            end_idx = min(start_idx + self.batch_size, n)
            yield RowBatch(self._csr_tr, self._csr_te, rows[start_idx:end_idx])

    def __iter__(self):
        if not self.resident:
            yield from self._iter_host()
            return
        for rb in self.iter_rows():
            data_tr = self._csr_tr.gather_dense(rb.rows)
            data_tr._rtx_rows = rb          # lets the trainer skip the dense detour when it gets this tensor back
            data_te = None
            if self._csr_te is not None:
                data_te = self._csr_te.gather_dense(rb.rows)
                data_te._rtx_rows = rb
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
