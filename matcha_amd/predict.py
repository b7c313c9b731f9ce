"""Inference consumers of a trained classifier on the HIP forward.

* ``predict_multiway`` -- the flow of the reference's ``predict_multiway.py``: parse a text file of multi-way
  interactions (one per line, tab-separated ``chrom:position`` items), map positions to node ids through
  ``temp_dir/bin2node.npy``, score every hyperedge with ``model(x)`` in chunks of 10 000 rows zero-padded PER CHUNK
  (predict_multiway.py:74-87 -- a row's logit depends on the chunk's width because pads are attended, SURVEY.md headline
  fact 7, so the chunking is part of the result), sigmoid, ``np.savetxt``.
* ``pairwise_probabilities`` / ``proba2matrix`` -- the pairwise sweep of ``denoise_contact.py``: all intra-chromosome
  pairs (i, j >= i + min_distance) scored at k = 2 (:67-74, :147-153) and scattered into a symmetric matrix (:32-62).
  Pairs are generated and scored on the device (rows of one width are independent, so the 10 000-row chunks of the
  reference do not matter here); the .mcool writer, the coverage normalisation and the plots stay out of scope.

CLI:  python -m matcha_amd.predict multiway -i interactions.txt -o output.txt
      python -m matcha_amd.predict pairwise --chrom 0 -o chr1_proba.npy
(both read ./config.JSON like the reference: temp_dir, resolution, chrom_list, min_distance).
"""
from __future__ import annotations

import argparse
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import utils as U

CHUNK_ROWS = 10000            # predict_multiway.py:77, denoise_contact.py:79


def parse_file(filepath: str, bin2node: Dict[str, int], chrom_list: Sequence[str], res: int) -> List[List[int]]:
    """predict_multiway.py:24-59: items of unknown chromosomes are skipped, positions are floored to their bin, node ids
    are de-duplicated and sorted, lines with fewer than two nodes are dropped.  An item without ``:`` raises EOFError and
    an unknown bin raises KeyError, as in the reference."""
    final = []
    with open(filepath, "r") as f:
        for line in f:
            temp = []
            for info in line.strip().split("\t"):
                try:
                    chrom, bin_ = info.split(":")
                except ValueError:
                    raise EOFError(info)
                if chrom not in chrom_list:
                    continue
                b = int(math.floor(int(bin_) / res)) * res
                temp.append(bin2node["%s:%d" % (chrom, b)])
            temp = sorted(set(temp))
            if len(temp) > 1:
                final.append(temp)
    return final


def predict(model, samples, batch_size: int = CHUNK_ROWS) -> np.ndarray:
    """Logits [n, 1] (numpy) of a list / array of hyperedges: eval mode, no grad, chunks of ``batch_size`` rows, each chunk
    zero-padded to ITS longest row (predict_multiway.py:74-87 == denoise_contact.py:76-88)."""
    model.eval()
    dev = model.layer_norm1.weight.device
    out = []
    with torch.no_grad():
        for j in range(0, len(samples), batch_size):
            x = U.pad_rows(samples[j:j + batch_size]).to(dev)
            out.append(model(x).detach().cpu().numpy())
    if not out:
        return np.zeros((0, 1), dtype=np.float32)
    return np.concatenate(out, axis=0)


def predict_multiway(model, filepath: str, bin2node: Dict[str, int], chrom_list: Sequence[str], res: int,
                     output: Optional[str] = None) -> Tuple[List[List[int]], np.ndarray]:
    """predict_multiway.py:104-113: parse, score, sigmoid, optionally ``np.savetxt`` (one probability per line)."""
    samples = parse_file(filepath, bin2node, chrom_list, res)
    proba = torch.sigmoid(torch.from_numpy(predict(model, samples))).numpy()
    if output is not None:
        np.savetxt(output, proba)
    return samples, proba


def generate_pair_wise(chrom_range, chrom_id: int, min_dis: int, device=None) -> torch.Tensor:
    """All pairs (i, j) with start <= i, i + min_dis <= j < end of one chromosome, in the reference's order
    (denoise_contact.py:67-74), as an int64 [n, 2] tensor built on ``device`` (no host loop)."""
    lo, hi = int(chrom_range[chrom_id][0]), int(chrom_range[chrom_id][1])
    i = torch.arange(lo, hi, dtype=torch.int64, device=device)
    cnt = torch.clamp(hi - i - int(min_dis), min=0)
    first = torch.repeat_interleave(i, cnt)
    start = torch.cumsum(cnt, 0) - cnt                                   # offset of each i's run
    j = torch.arange(int(cnt.sum()), dtype=torch.int64, device=device) - torch.repeat_interleave(start, cnt) + first + int(min_dis)
    return torch.stack([first, j], dim=1)


def pairwise_probabilities(model, chrom_range, chrom_id: int, min_dis: int, batch_rows: int = 1 << 20) -> Tuple[torch.Tensor, torch.Tensor]:
    """(pairs int64 [n, 2], probabilities float32 [n]) on the model's device: sigmoid(model(pairs)) at width L = 2
    (denoise_contact.py:147-153)."""
    model.eval()
    dev = model.layer_norm1.weight.device
    pairs = generate_pair_wise(chrom_range, chrom_id, min_dis, dev)
    out = torch.empty(len(pairs), dtype=torch.float32, device=dev)
    # back-to-back forwards: the per-call read-back of the device status word (one synchronisation each) is switched off and the
    # word is checked ONCE after the sweep (ids outside the model's tables are flagged on the device either way)
    check_each = getattr(model, "check_ids", None)
    if check_each is not None:
        model.check_ids = False
    try:
        with torch.no_grad():
            for s in range(0, len(pairs), batch_rows):
                out[s:s + batch_rows] = torch.sigmoid(model(pairs[s:s + batch_rows].contiguous()).reshape(-1))
        if check_each is not None:
            model.check_status()            # IndexError if chrom_range does not belong to this model (ids beyond its tables)
    finally:
        if check_each is not None:
            model.check_ids = check_each
    return pairs, out


def proba2matrix(sample, weight=None, proba=None, intra: bool = True):
    """denoise_contact.py:32-62 for numpy arrays or torch tensors (the matrix is built where ``sample`` lives).
    intra: symmetric [size, size] with m[i, j] (+)= p for every pair of columns of ``sample`` (indices relative to the
    smallest id), then m + m.T; inter: [size1, size2] of the two columns.  ``weight``: p -> max(p * weight, p).
    Like the reference, repeated (i, j) do not accumulate (fancy-index ``+=``) and the caller's ``sample`` is untouched
    (the reference shifts its argument in place)."""
    is_np = isinstance(sample, np.ndarray)
    s = torch.as_tensor(sample).long()
    p = torch.as_tensor(proba, device=s.device).float()
    if weight is not None:
        p = torch.maximum(p * torch.as_tensor(weight, device=s.device).float(), p)
    if intra:
        s = s - s.min()
        size = int(s.max()) + 1
        m = torch.zeros(size, size, dtype=torch.float32, device=s.device)
        for i in range(s.shape[-1] - 1):
            for j in range(i + 1, s.shape[-1]):
                m[s[:, i], s[:, j]] = m[s[:, i], s[:, j]] + p
        m = m + m.T
    else:
        a, b = s[:, 0] - s[:, 0].min(), s[:, 1] - s[:, 1].min()
        m = torch.zeros(int(a.max()) + 1, int(b.max()) + 1, dtype=torch.float32, device=s.device)
        m[a, b] = m[a, b] + p
    return m.cpu().numpy() if is_np else m


def _load(config_path: str = "./config.JSON"):
    import json
    with open(config_path) as f:
        config = json.load(f)
    temp_dir = config["temp_dir"]
    model = torch.load(os.path.join(temp_dir, "model2load"), map_location="cuda", weights_only=False)   # main.py:322, :685
    return config, temp_dir, model


def main(argv=None):
    ap = argparse.ArgumentParser(description="inference consumers of a trained MATCHA classifier on the MI355X path")
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("multiway", help="predict_multiway.py: probabilities of the multi-way interactions of a text file")
    a.add_argument("-i", "--file", type=str, required=True)
    a.add_argument("-o", "--output", type=str, default="./output.txt")
    b = sub.add_parser("pairwise", help="denoise_contact.py's sweep: probability matrix of all intra-chromosome pairs")
    b.add_argument("--chrom", type=int, required=True, help="index into config chrom_list")
    b.add_argument("-o", "--output", type=str, default="./pairwise.npy")
    for q in (a, b):
        q.add_argument("--config", type=str, default="./config.JSON")
    args = ap.parse_args(argv)
    config, temp_dir, model = _load(args.config)
    if args.cmd == "multiway":
        bin2node = np.load(os.path.join(temp_dir, "bin2node.npy"), allow_pickle=True).item()
        samples, proba = predict_multiway(model, args.file, bin2node, config["chrom_list"], config["resolution"], args.output)
        print("%d interactions -> %s" % (len(samples), args.output))
    else:
        chrom_range = np.load(os.path.join(temp_dir, "chrom_range.npy"))
        pairs, proba = pairwise_probabilities(model, chrom_range, args.chrom, config["min_distance"])
        np.save(args.output, proba2matrix(pairs, None, proba).cpu().numpy())
        print("%d pairs -> %s" % (len(pairs), args.output))


if __name__ == "__main__":
    main()
