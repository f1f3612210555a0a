r"""Readers of the reference's on-disk data formats (rectorch/data.py:275-557).

``rectorch.data.DataProcessing.process`` (data.py:89-219) leaves a pre-processed data set as a directory of

* ``train.csv``, ``validation_tr.csv``, ``validation_te.csv``, ``test_tr.csv``, ``test_te.csv`` -- header ``uid,iid`` (+ the
  rating and any further column, e.g. ``timestamp``, when the data set is not top-N), users and items as inner ids,
* ``unique_iid.txt`` / ``unique_uid.txt`` -- the raw ids in inner-id order.

:class:`DataReader` and :class:`DatasetManager` read that directory back into the ``scipy.sparse.csr_matrix`` objects
(or user -> item-sequence dictionaries for SVAE) that the samplers of this package upload to the MI355X, with the
reference's shapes, row order and dtype -- so a run pre-processed with rectorch moves onto this path unchanged.  The
pre-processing itself (raw ratings -> these files) is outside the hot path and not rebuilt here.

Host-side file parsing only: no compute path lives in this module.
"""
import json
import os

import numpy as np
import pandas as pd
from scipy import sparse

__all__ = ['DataReader', 'DatasetManager']


class _Cfg(dict):
    """attribute access to the data configuration keys, ``None`` for a missing key (the reference's DefaultMunch)"""
    def __getattr__(self, k):
        return self.get(k)


def _as_cfg(data_config):
    if isinstance(data_config, str):
        with open(data_config, "r") as f:
            return _Cfg(json.load(f))
    if isinstance(data_config, dict):
        return _Cfg(data_config)
    if hasattr(data_config, "proc_path"):
        return data_config
    raise TypeError("'data_config' must be of type 'DataConfig' (anything with .proc_path / .topn), dict or 'str'.")


class DataReader():
    r"""Reader of a pre-processed data set (reference data.py:275-495).

    Parameters
    ----------
    data_config : :obj:`str`, :obj:`dict` or an object with attributes ``proc_path``, ``topn`` (``seed``, ``test_prop``)
        The data configuration: the path of the reference's data-configuration ``.json`` file, its content, or the
        reference's own ``DataConfig`` object.

    Attributes
    ----------
    cfg
        The configuration.
    n_items : :obj:`int`
        The number of items in the data set (lines of ``unique_iid.txt``).
    """
    def __init__(self, data_config):
        self.cfg = _as_cfg(data_config)
        self.n_items = self._load_n_items()

    def _path(self, name):
        return os.path.join(self.cfg.proc_path, name)

    def _load_n_items(self):
        with open(self._path('unique_iid.txt'), 'r') as f:
            return sum(1 for _ in f)

    def _matrix(self, frame, n_rows, first_uid=0):
        rows = frame['uid'].to_numpy() - first_uid
        cols = frame['iid'].to_numpy()
        if self.cfg.topn:
            values = np.ones(len(frame))
        else:
            values = frame[frame.columns.values[2]].to_numpy()
        return sparse.csr_matrix((values, (rows, cols)), dtype='float64', shape=(n_rows, self.n_items))

    def load_data(self, datatype='train'):
        r"""Load (part of) the pre-processed data set (reference data.py:312-354).

        ``'train'``: the training matrix, one row per training user.  ``'validation'`` / ``'test'``: the pair (training
        part, test part) of those users, rows = the users of the block that have a non-empty training part, in inner-id
        order.  ``'full'``: training rows, then validation (tr + te), then test (tr + te) rows stacked.

        Raises
        ------
        :class:`ValueError`
            Raised when ``datatype`` does not match any of the valid strings.
        """
        if datatype == 'train':
            data = pd.read_csv(self._path('train.csv'))
            return self._matrix(data, int(data['uid'].max()) + 1)
        if datatype in ('validation', 'test'):
            data_tr = pd.read_csv(self._path('%s_tr.csv' % datatype))
            data_te = pd.read_csv(self._path('%s_te.csv' % datatype))
            first = int(min(data_tr['uid'].min(), data_te['uid'].min()))
            last = int(max(data_tr['uid'].max(), data_te['uid'].max()))
            if not self.cfg.topn:
                # the reference takes the rating column NAME from the training part for both files (data.py:397-398)
                data_te = data_te.rename(columns={data_te.columns.values[2]: data_tr.columns.values[2]})
            m_tr = self._matrix(data_tr, last - first + 1, first)
            m_te = self._matrix(data_te, last - first + 1, first)
            keep = np.diff(m_tr.indptr) != 0
            return m_tr[keep], m_te[keep]
        if datatype == 'full':
            tr = self.load_data('train')
            val_tr, val_te = self.load_data('validation')
            te_tr, te_te = self.load_data('test')
            return sparse.vstack([tr, val_tr + val_te, te_tr + te_te])
        raise ValueError("Possible datatype values are 'train', 'validation', 'test', 'full'.")

    # ---- user -> item-sequence dictionaries (the SVAE samplers' input) ----------------------------------------------
    @staticmethod
    def _to_dict(data, col="timestamp"):
        """user (re-based to 0) -> list of item ids ordered by ``col`` (stable for ties in file order after the global
        sort, as the reference's double sort is; data.py:411-418)"""
        data = data.sort_values(col, kind="stable")
        first = data["uid"].min()
        out = {}
        for uid, group in data.groupby("uid", sort=True):
            out[int(uid - first)] = list(group.sort_values(col, kind="stable")["iid"])
        return out

    def _split_by_time(self, data, col):
        """per user: the last ``max(int(test_prop * n), 1)`` items by ``col`` are the test part (data.py:420-441)"""
        test_prop = float(self.cfg.test_prop) if self.cfg.test_prop else 0.2
        tr, te = [], []
        for _, group in data.groupby(data.columns.values[0], sort=True):
            group = group.sort_values(col, kind="stable")
            sz = max(int(test_prop * len(group)), 1)
            tr.append(group.iloc[:len(group) - sz])
            te.append(group.iloc[len(group) - sz:])
        return pd.concat(tr), pd.concat(te)

    def load_data_as_dict(self, datatype='train', col="timestamp"):
        r"""Load the data as dictionaries user -> list of items sorted by ``col`` (reference data.py:443-495).
        ``'train'`` and ``'full'`` return one dictionary, ``'validation'`` / ``'test'`` the pair (training part, test
        part) obtained by re-splitting the block's interactions in time order."""
        if datatype == 'train':
            return self._to_dict(pd.read_csv(self._path('train.csv')), col)
        if datatype == 'full':
            names = ['train.csv', 'validation_tr.csv', 'validation_te.csv', 'test_tr.csv', 'test_te.csv']
            return self._to_dict(pd.concat([pd.read_csv(self._path(n)) for n in names]), col)
        if datatype not in ('validation', 'test'):
            raise ValueError("Possible datatype values are 'train', 'validation', 'test', 'full'.")
        combined = pd.concat([pd.read_csv(self._path('%s_tr.csv' % datatype)), pd.read_csv(self._path('%s_te.csv' % datatype))],
                             ignore_index=True)
        combined = combined.sort_values(col, kind="stable")
        data_tr, data_te = self._split_by_time(combined, col)
        return self._to_dict(data_tr, col), self._to_dict(data_te, col)


class DatasetManager():
    """Training, validation and test sets of a pre-processed data set (reference data.py:498-557).

    Attributes
    ----------
    n_items : :obj:`int`
    training_set : ``(csr_matrix, None)``
    validation_set, test_set : ``(csr_matrix, csr_matrix)``
        Training part and test part of the validation / test users.
    """
    def __init__(self, config_file):
        reader = DataReader(config_file)
        self.n_items = reader.n_items
        self.training_set = (reader.load_data('train'), None)
        self.validation_set = reader.load_data('validation')
        self.test_set = reader.load_data('test')

    def get_train_and_test(self):
        r"""Training + validation + the training part of the test users as one training matrix, and a test matrix of
        the same shape whose last rows are the test part of the test users (reference data.py:541-557)."""
        tr = sparse.vstack([self.training_set[0], self.validation_set[0] + self.validation_set[1], self.test_set[0]])
        n_pad = tr.shape[0] - self.test_set[1].shape[0]
        te = sparse.vstack([sparse.csr_matrix((n_pad, tr.shape[1])), self.test_set[1]])
        return tr, te
