r"""Evaluation utilities with the reference's API (rectorch/evaluation.py:11-178).

These are the host-side consumers of the hot path: they iterate a sampler, call ``model.predict`` (the HIP
forward) and score the returned logits with :class:`rectorch_amd.metrics.Metrics`.  The duck-typed contract
is the reference's: ``predict(x)[0]`` supports ``.cpu().numpy()``, samplers yield ``(data_tr, heldout)``
pairs supporting ``.view``, ``.shape`` and ``.cpu().numpy()`` (evaluation.py:100-103).
"""
from functools import partial
import inspect
import random

import numpy as np
import torch

from .metrics import Metrics

__all__ = ['ValidFunc', 'evaluate', 'evaluate_host', 'evaluate_device', 'one_plus_random']

DEVICE_TOPK_MAX = 1024


class ValidFunc():
    """Wrapper adapting an evaluation function to the ``(model, test_loader, metric_list)`` signature the
    trainers call (reference evaluation.py:11-64): extra keyword arguments are bound at construction and the
    remaining positional arguments must be exactly those three names.

    >>> opr = ValidFunc(one_plus_random, r=5)
    >>> opr
    ValidFunc(fun='one_plus_random', params={'r': 5})
    """
    def __init__(self, func, **kwargs):
        self.func_name = func.__name__
        self.function = partial(func, **kwargs)
        args = inspect.getfullargspec(self.function).args
        assert args == ["model", "test_loader", "metric_list"],\
            "A (partial) validation function must have the following kwargs: model, test_loader and\
            metric_list"

    def __call__(self, model, test_loader, metric):
        return self.function(model, test_loader, [metric])[metric]

    def __str__(self):
        kwdefargs = inspect.getfullargspec(self.function).kwonlydefaults
        return "ValidFunc(fun='%s', params=%s)" % (self.func_name, kwdefargs)

    def __repr__(self):
        return str(self)


def _to_numpy(t):
    from .engine import RowBatch
    if isinstance(t, RowBatch):            # rows of a device-resident CSR (sparse samplers): densify on the device
        return t.tr.gather_dense(t.rows).cpu().numpy()
    t = t.view(t.shape[0], -1)
    return t.cpu().numpy()


class _PerUserResults:
    """per-metric lists of per-batch arrays -> one array per metric, in loader order (reference evaluation.py:104-109)"""
    def __init__(self, metric_list):
        self._parts = {m: [] for m in metric_list}

    def add(self, batch_result):
        for m, values in batch_result.items():
            self._parts[m].append(values)

    def finish(self):
        return {m: np.concatenate(parts) for m, parts in self._parts.items()}


def _predict_numpy(model, data_tr):
    """scores of one batch as a host array; the resident-rows shortcut of the device sampler survives the reshape"""
    from .engine import RowBatch, tag_rows, tagged_rows
    if isinstance(data_tr, RowBatch):
        return model.predict(data_tr)[0].cpu().numpy()
    data_tensor = data_tr.view(data_tr.shape[0], -1)
    rows = tagged_rows(data_tr)
    if rows is not None:
        tag_rows(data_tensor, rows)
    return model.predict(data_tensor)[0].cpu().numpy()


def evaluate_host(model, test_loader, metric_list):
    r"""The reference's loop as it is written (evaluation.py:100-109): ``predict`` per batch, the ``[B, n_items]`` scores and the
    held-out rows copied to the host, :class:`Metrics` in numpy.  What :func:`evaluate` falls back to."""
    out = _PerUserResults(metric_list)
    for data_tr, heldout in test_loader:
        out.add(Metrics.compute(_predict_numpy(model, data_tr), _to_numpy(heldout), metric_list))
    return out.finish()


def evaluate(model, test_loader, metric_list):
    r"""Evaluate ``model`` on every batch of ``test_loader`` with every metric of ``metric_list``
    (``"name@k"`` strings).  Returns ``dict metric -> per-user numpy array`` in loader order
    (reference evaluation.py:67-110).

    Same signature, same values -- and since round 5 the same SPEED as :func:`evaluate_device` wherever that applies: a
    device-resident :class:`DataSampler` with held-out rows and ``ndcg@k`` / ``recall@k`` metrics (k <= 1024) are scored by the
    top-k kernel on the GPU (4.3 M users/s against 8 K through the host loop), so ``model.train(...)``'s default
    ``valid_func=ValidFunc(evaluate)`` no longer spends its time copying score matrices.  Everything else -- other metrics, host
    samplers, models without the device path, ``model.device_metrics = False`` -- takes the reference's loop
    (:func:`evaluate_host`); the two agree to 1e-12 (``test_evaluate_device_equals_host_evaluate``).  A subclass that overrides
    ``predict`` always takes the host loop (its override is what the reference would call).  Ties: among EQUAL scores the device
    top-k keeps the lower item index where numpy's argpartition order is unspecified; metrics differ only if a held-out item ties
    with a non-held-out one exactly at rank k."""
    if (getattr(model, "device_metrics", True) and _device_plan(test_loader, metric_list) is not None and hasattr(model, "_predict_tuple")
            and _predict_is_ours(model)):
        return evaluate_device(model, test_loader, metric_list)
    return evaluate_host(model, test_loader, metric_list)


def _predict_is_ours(model):
    """True when ``model.predict`` is the framework's own method.  The reference always scores through ``model.predict``
    (evaluation.py:100-109), so a user subclass that overrides it (re-ranking, filtering, an ensemble) must be evaluated through
    that override: the device route calls the engine's scorer directly and would silently bypass it."""
    fn = getattr(type(model), "predict", None)
    mod = getattr(fn, "__module__", "") or ""
    return "predict" not in vars(model) and mod.startswith(__name__.rsplit(".", 1)[0] + ".")


def _device_plan(test_loader, metric_list):
    """[(metric, name, k)] when every metric can be computed by the device kernel on this loader, else None"""
    from .samplers import DataSampler
    parsed = []
    for m in metric_list:
        name, _, k = m.partition("@")
        if name.lower() not in ("ndcg", "recall") or not k.isdigit() or not 1 <= int(k) <= DEVICE_TOPK_MAX:
            return None
        parsed.append((m, name.lower(), int(k)))
    resident = isinstance(test_loader, DataSampler) and test_loader.resident and test_loader.sparse_data_te is not None
    return parsed if (parsed and resident) else None


def evaluate_device(model, test_loader, metric_list):
    r"""Same contract and same values as :func:`evaluate`, computed on the MI355X.

    With a device-resident :class:`rectorch_amd.samplers.DataSampler` holding the ``(tr, heldout)`` matrices,
    ``predict`` takes the sparse rows directly and the ``ndcg@k`` / ``recall@k`` metrics are computed by a top-k
    kernel on the GPU, so per batch only ``len(metric_list) x B`` doubles cross PCIe instead of the ``[B, n_items]``
    score matrix (40 MB per 500 users at the ml-20m shape) followed by a host ``argpartition``.  Usable as a
    validation function: ``model.train(..., valid_func=ValidFunc(evaluate_device))``.  Anything it cannot do on the
    device (other metrics, k > 1024, a host sampler) goes through :func:`evaluate`.
    """
    from .engine import topk_metrics, RowBatch
    parsed = _device_plan(test_loader, metric_list)
    if parsed is None:
        return evaluate_host(model, test_loader, metric_list)
    ks = sorted({k for _, _, k in parsed})
    out = _PerUserResults(metric_list)
    # Round 6, after the selection kernel went from 41 to 20 us per 500 users: the GPU is busy ~103 us per batch
    # (profiles/r6_eval_timeline.txt) and the host needed as long to get through one iteration of a Python loop (predict -> ctypes ->
    # eight launches, the selection kernel's call, two allocations).  When `predict` is the framework's own (a user's override must be
    # what scores, as in the reference) the WHOLE loop is one C call -- rtx_engine_evaluate_topk enqueues forward, -inf scatter and
    # selection kernel batch after batch into one scores buffer and one [cut-off][user] metrics buffer -- and ONE device -> host copy
    # follows.  (Measured and dropped earlier this round: the selection kernel on a second stream under the next forward -- both
    # contend for the same CUs; and the Python loop with reused buffers: +-1 %.)
    batches = list(test_loader.iter_rows())
    if not batches:
        return out.finish()
    fast = (_predict_is_ours(model) and hasattr(model, "_predict_engine") and getattr(model, "_variant", None) in ("vae", "dae")
            and all(isinstance(model.network._as_input(rb), RowBatch) and rb.tr is batches[0].tr and rb.te is batches[0].te for rb in batches))
    import os
    if os.environ.get("RTX_EVAL_SIMPLE_LOOP"): fast = False          # (measurement: the per-batch Python loop)
    if fast:
        eng = model._predict_engine(max(len(rb) for rb in batches))
        offsets = np.concatenate([[0], np.cumsum([len(rb) for rb in batches])])
        base, off0 = batches[0].rows._base, batches[0].rows.storage_offset()
        if (base is not None and base.dim() == 1 and base.is_contiguous()
                and all(rb.rows._base is base and rb.rows.storage_offset() == off0 + int(o) for rb, o in zip(batches, offsets))):
            rows = base[off0:off0 + int(offsets[-1])]        # the sampler's batches are consecutive slices of ONE row-number tensor
        else:
            rows = torch.cat([rb.rows for rb in batches])
        dn, dr = eng.evaluate_topk(batches[0].tr, batches[0].te, rows, offsets, ks)
        ndcg, recall = dn.cpu().numpy(), dr.cpu().numpy()
    else:
        per_batch = []
        for rb in batches:
            scores = model.predict(rb)[0]                # HIP forward on the sparse rows, -inf at the train items
            per_batch.append(topk_metrics(scores, rb.te, rb.rows, ks))
        # ONE device -> host copy for the whole loader (the per-batch .cpu() of round 3 was a host sync per 500 users)
        ndcg = torch.cat([n for n, _ in per_batch], dim=1).cpu().numpy()
        recall = torch.cat([r for _, r in per_batch], dim=1).cpu().numpy()
    out.add({m: (ndcg if name == "ndcg" else recall)[ks.index(k)] for m, name, k in parsed})
    return out.finish()


def one_plus_random(model, test_loader, metric_list, r=1000):
    r"""One-plus-random evaluation (reference evaluation.py:113-178): for every held-out positive of every
    user, rank it against ``r`` random items the user has not interacted with in the held-out part and
    compute the metrics on those ``r + 1`` scores (the positive is column 0).  Raises ``ValueError`` when
    fewer than ``r`` negatives exist."""
    out = _PerUserResults(metric_list)
    for data_tr, heldout in test_loader:
        scores = _predict_numpy(model, data_tr)
        heldout = _to_numpy(heldout)
        all_items = set(range(heldout.shape[1]))
        contests = []
        for u, i in zip(*heldout.nonzero()):                 # one contest per held-out positive, in row-major order
            negatives = sorted(all_items - set(heldout[u].nonzero()[0].tolist()))
            contests.append(scores[u][[i] + random.sample(negatives, r)])   # the same draws as the reference's sampler
        pred = np.array(contests)
        truth = np.zeros_like(pred)
        truth[:, 0] = 1
        out.add(Metrics.compute(pred, truth, metric_list))
    return out.finish()
