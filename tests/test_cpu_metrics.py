"""The per-epoch metrics of the driver (reference utils.py:32-72: sklearn's roc_auc_score / average_precision_score per hyperedge
size, accuracy per size) computed with a sort and prefix sums on the tensors' own device (matcha_amd/utils.py) against scikit-learn
itself: raw values to 1e-12, the reference's formatted strings equal -- ties, one-class subsets and the reference's (0.0, 0.0)
failure value included."""
import numpy as np
import pytest
import torch

from matcha_amd import utils as U


def _sklearn_strings(y, p, s):
    from sklearn.metrics import average_precision_score, roc_auc_score
    yt = (y > 0.5).astype(np.float64)
    roc = "%s %.3f " % ("all", roc_auc_score(yt, p))
    pr = "%s %.3f " % ("all", average_precision_score(yt, p))
    for k in np.unique(s):
        m = s == k
        roc += "%s %.3f " % (str(k), roc_auc_score(yt[m], p[m]))
        pr += "%s %.3f " % (str(k), average_precision_score(yt[m], p[m]))
    return roc[:-1], pr[:-1]


@pytest.mark.parametrize("n,levels,seed", [(50, 0, 0), (5000, 0, 1), (5000, 7, 2), (200000, 1000, 3), (3000, 2, 4)])
def test_auc_and_average_precision_equal_sklearn(n, levels, seed):
    """levels > 0: scores quantised to that many values (heavy ties: the thresholds are the DISTINCT scores)."""
    from sklearn.metrics import average_precision_score, roc_auc_score
    rng = np.random.default_rng(seed)
    y = (rng.random(n) < 0.3).astype(np.float32)
    p = np.clip(0.35 * y + rng.random(n) * 0.8, 0, 1).astype(np.float32)
    if levels:
        p = (np.floor(p * levels) / levels).astype(np.float32)
    s = rng.integers(2, 6, size=n)
    auc, ap = U._auc_ap(torch.from_numpy(y) > 0.5, torch.from_numpy(p))
    assert abs(auc - roc_auc_score(y, p)) < 1e-12
    assert abs(ap - average_precision_score(y, p)) < 1e-12
    got = U.roc_auc_cuda(torch.from_numpy(y), torch.from_numpy(p), torch.from_numpy(s), 5)
    assert got == _sklearn_strings(y, p, s)
    acc = U.accuracy(torch.from_numpy(p), torch.from_numpy(y), torch.from_numpy(s))
    want = "".join("%s %.3f " % (str(k), float(((p[s == k] >= 0.5) == (y[s == k] >= 0.5)).mean())) for k in np.unique(s))
    assert acc == want


def test_one_class_subset_gives_the_reference_failure_value():
    """sklearn raises for a subset with one class only; the reference catches everything and returns (0.0, 0.0) (utils.py:53-54)."""
    y = torch.tensor([1., 0., 1., 1.])
    p = torch.tensor([.9, .2, .6, .4])
    s = torch.tensor([2, 2, 3, 3])                     # size 3 holds positives only
    assert U.roc_auc_cuda(y, p, s, 3) == (0.0, 0.0)
    with pytest.raises(ValueError):
        U._auc_ap(torch.tensor([True, True]), torch.tensor([0.1, 0.2]))


def test_non_finite_scores_give_the_reference_failure_value():
    """sklearn refuses NaN / inf scores, so the reference's `except` logs (0.0, 0.0) for such an epoch (utils.py:32-54): a diverged run
    must not come back looking like a score."""
    import torch
    from matcha_amd.utils import roc_auc_cuda
    y = torch.tensor([1.0, 0.0, 1.0, 0.0, 1.0, 0.0])
    sz = torch.tensor([2, 2, 3, 3, 2, 3])
    for bad in (float("nan"), float("inf")):
        p = torch.tensor([0.9, 0.1, 0.8, bad, 0.7, 0.3])
        assert roc_auc_cuda(y, p, sz, 3) == (0.0, 0.0)
