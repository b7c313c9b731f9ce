"""Host-side helpers on the hot path's edges, mirroring the pieces of the reference's ``Code/utils.py`` that
``main.py`` / ``predict_multiway.py`` use (np2tensor_hyper :24-29, roc_auc_cuda :32-54, accuracy :57-72,
build_hash :75-97, sync_shuffle :142-149, get_config :157-159).  Off the step path: plain numpy / torch."""
from __future__ import annotations

import json
from typing import List, Sequence

import numpy as np
import torch


def get_config(path: str = "./config.JSON") -> dict:
    """The 13-key run configuration, read from the working directory like the reference does."""
    with open(path, "r") as f:
        return json.load(f)


def np2tensor_hyper(vec, dtype=torch.long):
    """Rectangular input -> one tensor; ragged input -> list of 1-D tensors (reference utils.py:24-29).  Unlike the
    reference this also works on numpy >= 1.24, where np.asarray of a ragged list raises."""
    try:
        arr = np.asarray(vec)
        if arr.dtype != object and arr.ndim == 2:
            return torch.as_tensor(arr, dtype=dtype)
    except ValueError:
        pass
    return [torch.as_tensor(np.asarray(v), dtype=dtype) for v in vec]


def pad_rows(rows) -> torch.Tensor:
    """Zero-pad hyperedges of mixed size to the widest one in the chunk (what pad_sequence(batch_first=True,
    padding_value=0) does at main.py:436-437 / predict_multiway.py:82)."""
    t = np2tensor_hyper(rows)
    if isinstance(t, list):
        return torch.nn.utils.rnn.pad_sequence(t, batch_first=True, padding_value=0)
    return t


def sync_shuffle(sample_list: Sequence, max_num: int = -1) -> List:
    """Same random permutation applied to every array of the list, optionally truncated (reference utils.py:142-149)."""
    index = torch.randperm(len(sample_list[0]))
    if max_num > 0:
        index = index[:max_num]
    return [s[index] for s in sample_list]


def _binary_counts(y: torch.Tensor, p: torch.Tensor):
    """Cumulative true / false positives at every DISTINCT score, scores descending -- what sklearn's ``_binary_clf_curve`` returns
    (stable sort, thresholds where the sorted score changes + the last one), on whatever device the tensors live on."""
    order = torch.argsort(p, descending=True, stable=True)
    ps, ys = p[order], y[order].double()
    n = ps.numel()
    last = torch.tensor([n - 1], device=ps.device)
    idx = torch.cat([torch.nonzero(ps[1:] != ps[:-1]).reshape(-1), last])
    tps = torch.cumsum(ys, 0)[idx]
    fps = (idx + 1).double() - tps
    return fps, tps


def _auc_ap(y: torch.Tensor, p: torch.Tensor):
    """(roc_auc_score, average_precision_score) of binary labels ``y`` (0 / 1) and scores ``p``, with sklearn's definitions:
    trapezoidal area under the ROC curve through (0, 0) and the points of the distinct thresholds (ties form one point);
    AP = sum_i (R_i - R_{i-1}) P_i over the distinct thresholds, R_{-1} = 0.  ValueError with one class only, like sklearn."""
    if y.numel() == 0:
        raise ValueError("no samples")
    if not bool(torch.isfinite(p).all()):
        # sklearn's input validation: the reference's `except` then logs (0.0, 0.0) for the epoch -- a diverged run must not look like a score
        raise ValueError("Input contains NaN or infinity.")
    fps, tps = _binary_counts(y, p)
    n_pos, n_neg = float(tps[-1]), float(fps[-1])
    if n_pos <= 0 or n_neg <= 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    zero = torch.zeros(1, dtype=torch.float64, device=fps.device)
    fpr, tpr = torch.cat([zero, fps / n_neg]), torch.cat([zero, tps / n_pos])
    auc = float(torch.sum((fpr[1:] - fpr[:-1]) * (tpr[1:] + tpr[:-1])) * 0.5)
    precision = tps / (tps + fps)
    recall = torch.cat([zero, tps / n_pos])
    ap = float(torch.sum((recall[1:] - recall[:-1]) * precision))
    return auc, ap


def roc_auc_cuda(y_true, y_pred, size_list, max_size):
    """'all <auc> <k> <auc> ...' and the same for average precision (reference utils.py:32-54); (0.0, 0.0) on failure.
    The reference hands numpy copies to scikit-learn once per epoch (0.6 s for the 1.5 M rows of an epoch at its own settings: 40 % of
    the epoch's wall clock here); this computes the same two numbers -- sklearn's definitions, ties included, float64 -- with a sort
    and two prefix sums on the tensors' own device (tests/test_cpu_metrics.py pins them against scikit-learn)."""
    try:
        yt = (torch.as_tensor(y_true) > 0.5).reshape(-1)
        yp = torch.as_tensor(y_pred).detach().reshape(-1).to(yt.device)
        sz = torch.as_tensor(size_list).reshape(-1).to(yt.device)
        auc, ap = _auc_ap(yt, yp)
        roc_s = "%s %.3f " % ("all", auc)
        pr_s = "%s %.3f " % ("all", ap)
        for s in torch.unique(sz).tolist():
            m = sz == s
            auc, ap = _auc_ap(yt[m], yp[m])
            roc_s += "%s %.3f " % (str(s), auc)
            pr_s += "%s %.3f " % (str(s), ap)
        return roc_s[:-1], pr_s[:-1]
    except BaseException:
        return 0.0, 0.0


def accuracy(output, target, size_list=None, max_size=None) -> str:
    """Fraction of rows on the right side of 0.5, per hyperedge size (reference utils.py:57-72); on the tensors' own device."""
    out = torch.as_tensor(output).detach().reshape(-1)
    tgt = torch.as_tensor(target).detach().reshape(-1).to(out.device)
    hit = ((out >= 0.5) == (tgt >= 0.5)).double()
    if size_list is None:
        return "%.3f " % float(hit.mean())
    sz = torch.as_tensor(size_list).reshape(-1).to(out.device)
    s_out = ""
    for s in torch.unique(sz).tolist():
        s_out += "%s %.3f " % (str(s), float(hit[sz == s].mean()))
    return s_out


def build_hash(data, compress=None, min_size=None, max_size=None, capacity=None, device="cuda"):
    """Membership structure over known hyperedges (reference utils.py:75-97 builds one Bloom filter per size; here:
    one exact device hash set for all sizes).  ``data``: rows of mixed size or a zero-padded int array."""
    from .sampler import HyperedgeSet
    x = pad_rows(data)
    return HyperedgeSet(x.to(device))
