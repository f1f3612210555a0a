r"""Ranking metrics with the reference's API and definitions (rectorch/metrics.py:31-285).

Host-side numpy, like the reference (which uses Bottleneck's ``argpartition``; numpy's is used here: the two can differ
only in the order inside the unsorted top-k set, which nDCG re-sorts and recall / hit treat as a set).  Pinned by the
reference's own known-answer tests and golden vector G6.  The device-side counterpart used by ``evaluate_device`` is
``rtx_topk_metrics`` (rectorch_amd/engine.py::topk_metrics).

All four metrics are views of one quantity, the relevance of the k best-scored items of every user:

* ``_top_relevance(..., ranked=True)``  -> ``[n_users, k]`` relevance in rank order (nDCG, MRR);
* ``_top_relevance(..., ranked=False)`` -> the same entries in arbitrary order (recall, hit: only their count matters).
"""
import logging

import numpy as np

__all__ = ['Metrics']

logger = logging.getLogger(__name__)


def _check(pred_scores, ground_truth, k):
    assert pred_scores.shape == ground_truth.shape,\
        "'pred_scores' and 'ground_truth' must have the same shape."
    return min(pred_scores.shape[1], k)


def _top_relevance(pred_scores, ground_truth, k, ranked):
    """relevance (entries of ``ground_truth``) of each user's k best-scored items"""
    users = np.arange(pred_scores.shape[0])[:, None]
    top = np.argpartition(-pred_scores, k - 1, axis=1)[:, :k]          # the top-k SET, O(n_items) per user
    if ranked:
        top = top[users, np.argsort(-pred_scores[users, top], axis=1)]   # ... put in rank order
    return ground_truth[users, top]


class Metrics:
    r"""Static metric functions; users on the rows, items on the columns."""

    @staticmethod
    def compute(pred_scores, ground_truth, metrics_list):
        r"""Evaluate every entry of ``metrics_list``: ``"name@k"`` calls ``name_at_k(pred_scores, ground_truth, k)``, a
        bare method name is called without ``k``, unknown names are skipped with a warning (reference metrics.py:74-85).
        Returns ``dict name -> per-user array``."""
        results = {}
        for spec in metrics_list:
            name, _, cut = spec.partition("@")
            fn = getattr(Metrics, "%s_at_k" % name.lower() if cut else name, None)
            if fn is None:
                logger.warning("Skipped unknown metric '%s'.", spec)
                continue
            results[spec] = fn(pred_scores, ground_truth, int(cut)) if cut else fn(pred_scores, ground_truth)
        return results

    @staticmethod
    def ndcg_at_k(pred_scores, ground_truth, k=100):
        r"""nDCG@k with binary relevance: discounted gain of the k best-scored items over the ideal one, the ideal list
        holding ``min(k, #relevant)`` hits (reference metrics.py:136-147)."""
        k = _check(pred_scores, ground_truth, k)
        discount = 1. / np.log2(np.arange(2, k + 2))
        gain = (_top_relevance(pred_scores, ground_truth, k, ranked=True) * discount).sum(axis=1)
        ideal_cum = np.concatenate(([0.], np.cumsum(discount)))
        n_rel = np.minimum(ground_truth.sum(axis=1).astype(np.int64), k)
        return gain / ideal_cum[n_rel]

    @staticmethod
    def recall_at_k(pred_scores, ground_truth, k=100):
        r"""Recall@k normalised by ``min(k, #relevant)`` (reference metrics.py:187-196)."""
        k = _check(pred_scores, ground_truth, k)
        hits = (_top_relevance(pred_scores, ground_truth, k, ranked=False) > 0).sum(axis=1).astype(np.float32)
        return hits / np.minimum(k, (ground_truth > 0).sum(axis=1))

    @staticmethod
    def hit_at_k(pred_scores, ground_truth, k=100):
        r"""Whether any relevant item is among the k best-scored (reference metrics.py:231-238)."""
        k = _check(pred_scores, ground_truth, k)
        return (_top_relevance(pred_scores, ground_truth, k, ranked=False) > 0).any(axis=1)

    @staticmethod
    def mrr_at_k(pred_scores, ground_truth, k=100):
        r"""Reciprocal rank of the first relevant item within the top k, 0 if there is none (reference
        metrics.py:272-285)."""
        k = _check(pred_scores, ground_truth, k)
        rel = _top_relevance(pred_scores, ground_truth, k, ranked=True) != 0
        first = np.argmax(rel, axis=1)                                   # 0 when there is no hit: masked below
        return np.where(rel.any(axis=1), 1. / (1. + first), 0.)
