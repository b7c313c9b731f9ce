"""Host-side helpers on the hot path's edges, mirroring the pieces of the reference's ``Code/utils.py`` that
``main.py`` / ``predict_multiway.py`` use (np2tensor_hyper :24-29, roc_auc_cuda :32-54, accuracy :57-72,
build_hash :75-97, sync_shuffle :142-149, get_config :157-159).  Off the step path: plain numpy / torch / sklearn."""
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


def roc_auc_cuda(y_true, y_pred, size_list, max_size):
    """'all <auc> <k> <auc> ...' and the same for average precision (reference utils.py:32-54); (0.0, 0.0) on failure."""
    from sklearn.metrics import average_precision_score, roc_auc_score
    try:
        yt = (torch.as_tensor(y_true) > 0.5).float().cpu().numpy().reshape(-1)
        yp = torch.as_tensor(y_pred).detach().cpu().numpy().reshape(-1)
        sz = np.asarray(torch.as_tensor(size_list).cpu()).reshape(-1)
        roc_s = "%s %.3f " % ("all", roc_auc_score(yt, yp))
        pr_s = "%s %.3f " % ("all", average_precision_score(yt, yp))
        for s in np.unique(sz):
            m = sz == s
            roc_s += "%s %.3f " % (str(s), roc_auc_score(yt[m], yp[m]))
            pr_s += "%s %.3f " % (str(s), average_precision_score(yt[m], yp[m]))
        return roc_s[:-1], pr_s[:-1]
    except BaseException:
        return 0.0, 0.0


def accuracy(output, target, size_list=None, max_size=None) -> str:
    """Fraction of rows on the right side of 0.5, per hyperedge size (reference utils.py:57-72)."""
    out = torch.as_tensor(output).detach().cpu().reshape(-1)
    tgt = torch.as_tensor(target).detach().cpu().reshape(-1)
    if size_list is None:
        return "%.3f " % float(((out >= 0.5) == (tgt >= 0.5)).float().mean())
    sz = torch.as_tensor(size_list).cpu().reshape(-1)
    s_out = ""
    for s in torch.unique(sz).tolist():
        m = sz == s
        s_out += "%s %.3f " % (str(s), float(((out[m] >= 0.5) == (tgt[m] >= 0.5)).float().mean()))
    return s_out


def build_hash(data, compress=None, min_size=None, max_size=None, capacity=None, device="cuda"):
    """Membership structure over known hyperedges (reference utils.py:75-97 builds one Bloom filter per size; here:
    one exact device hash set for all sizes).  ``data``: rows of mixed size or a zero-padded int array."""
    from .sampler import HyperedgeSet
    x = pad_rows(data)
    return HyperedgeSet(x.to(device))
