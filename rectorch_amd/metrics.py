r"""Ranking metrics with the reference's API and definitions (rectorch/metrics.py:31-285).

Host-side numpy, like the reference (which uses Bottleneck's ``argpartition``; numpy's is used here: the two
can differ only in the order inside the unsorted top-k set, which ``ndcg_at_k`` re-sorts and ``recall_at_k``
treats as a set).  Pinned by the reference's own known-answer tests and golden vector G6.
"""
import logging

import numpy as np

__all__ = ['Metrics']

logger = logging.getLogger(__name__)


def _topk_idx(pred_scores, k):
    """indices of the k largest scores per row, unordered (argpartition on -scores, as the reference)."""
    return np.argpartition(-pred_scores, k - 1, axis=1)[:, :k]


class Metrics:
    r"""Static metric functions; users on the rows, items on the columns."""

    @staticmethod
    def compute(pred_scores, ground_truth, metrics_list):
        r"""Compute every metric in ``metrics_list`` (strings ``"name@k"`` -> ``name_at_k(..., k)``; a bare
        method name is called without k; unknown names are skipped with a warning) -- metrics.py:74-85."""
        results = {}
        for metric in metrics_list:
            try:
                if "@" in metric:
                    met, k = metric.split("@")
                    met_foo = getattr(Metrics, "%s_at_k" % met.lower())
                    results[metric] = met_foo(pred_scores, ground_truth, int(k))
                else:
                    results[metric] = getattr(Metrics, metric)(pred_scores, ground_truth)
            except AttributeError:
                logger.warning("Skipped unknown metric '%s'.", metric)
        return results

    @staticmethod
    def ndcg_at_k(pred_scores, ground_truth, k=100):
        r"""nDCG@k with binary relevance: DCG over the k best-scored items / ideal DCG (metrics.py:136-147)."""
        assert pred_scores.shape == ground_truth.shape,\
            "'pred_scores' and 'ground_truth' must have the same shape."
        k = min(pred_scores.shape[1], k)
        n_users = pred_scores.shape[0]
        rows = np.arange(n_users)[:, np.newaxis]
        part = _topk_idx(pred_scores, k)
        order = np.argsort(-pred_scores[rows, part], axis=1)
        idx_topk = part[rows, order]
        tp = 1. / np.log2(np.arange(2, k + 2))
        DCG = (ground_truth[rows, idx_topk] * tp).sum(axis=1)
        IDCG = np.array([(tp[:min(int(n), k)]).sum() for n in ground_truth.sum(axis=1)])
        return DCG / IDCG

    @staticmethod
    def recall_at_k(pred_scores, ground_truth, k=100):
        r"""Recall@k normalised by min(k, #relevant) (metrics.py:187-196)."""
        assert pred_scores.shape == ground_truth.shape,\
            "'pred_scores' and 'ground_truth' must have the same shape."
        k = min(pred_scores.shape[1], k)
        rows = np.arange(pred_scores.shape[0])[:, np.newaxis]
        hit = np.zeros_like(pred_scores, dtype=bool)
        hit[rows, _topk_idx(pred_scores, k)] = True
        rel = (ground_truth > 0)
        num = (np.logical_and(rel, hit).sum(axis=1)).astype(np.float32)
        return num / np.minimum(k, rel.sum(axis=1))

    @staticmethod
    def hit_at_k(pred_scores, ground_truth, k=100):
        r"""Whether any relevant item is among the k best-scored (metrics.py:231-238)."""
        assert pred_scores.shape == ground_truth.shape,\
            "'pred_scores' and 'ground_truth' must have the same shape."
        k = min(pred_scores.shape[1], k)
        rows = np.arange(pred_scores.shape[0])[:, np.newaxis]
        hit = np.zeros_like(pred_scores, dtype=bool)
        hit[rows, _topk_idx(pred_scores, k)] = True
        num = (np.logical_and(ground_truth > 0, hit).sum(axis=1)).astype(np.float32)
        return num > 0

    @staticmethod
    def mrr_at_k(pred_scores, ground_truth, k=100):
        r"""Reciprocal rank of the first relevant item within the top k, 0 if none (metrics.py:272-285)."""
        assert pred_scores.shape == ground_truth.shape,\
                "'pred_scores' and 'ground_truth' must have the same shape."
        k = min(pred_scores.shape[1], k)
        idx = np.argsort(-pred_scores)
        hits = ground_truth[np.arange(ground_truth.shape[0])[:, np.newaxis], idx[:, :k]]
        mrr = np.zeros(ground_truth.shape[0])
        for r, c in zip(*hits.nonzero()):
            if mrr[r] == 0:
                mrr[r] = 1. / (1 + c)
        return mrr
