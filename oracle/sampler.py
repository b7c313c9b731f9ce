"""Oracle: negative sampling (reference main.py:361-459) restated in plain Python.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Two things are restated here:
  * the ACCEPT / REJECT RULES of the reference's ``generate_negative`` -- same size as the positive, replaced
    nodes stay in their chromosome, duplicate-free, sorted, min adjacent gap > min_dis, not a known hyperedge,
    negatives == positives when the positive itself is not in the set (the reference's phase 1, main.py:589);
  * this project's RANDOM STREAM for it (oracle/rng.py, STREAM_NEG), so that the HIP kernel
    (matcha_amd/csrc/sampler.hip) can be compared BIT FOR BIT with this file.

What cannot be pinned: the reference's own random stream (python ``random`` + numpy global)
and ``pybloom_live``'s false positives (third-party, un-vendored, absent here): PARITY UNPINNED at that boundary;
tests compare distributions with statistics captured from the reference (tests/golden/sampler_stats.npz).
"""
from __future__ import annotations

import numpy as np

from . import rng as R

MAX_TRIALS = 1 << 16


def _u32(key, hi, lo):
    return int(R.rand_u32(key, np.uint64(hi), np.uint64(lo)))


def sample_negatives(pos: np.ndarray, known: set, node2chrom: np.ndarray, chrom_range: np.ndarray, neg_num: int,
                     min_dis: int, seed: int) -> np.ndarray:
    """pos int64 [P,L] zero-padded ascending rows; ``known`` = set of tuples (without padding).
    Returns int64 [P*neg_num, L]; negatives of positive j are rows neg_num*j .. neg_num*j+neg_num-1 (main.py:383-428)."""
    pos = np.asarray(pos, dtype=np.int64)
    P, L = pos.shape
    out = np.zeros((P * neg_num, L), dtype=np.int64)
    key = R.make_key(seed, R.STREAM_NEG)
    for j in range(P):
        orig = [int(v) for v in pos[j] if v != 0]
        k = len(orig)
        for i in range(neg_num):
            n = j * neg_num + i
            result = list(orig)
            # `while neighbor_check(temp, dict)` is entered only if the positive is a member (main.py:390-392)
            if len(known) > 0 and k > 0 and tuple(orig) in known:
                mask, a = 0, 0
                while mask == 0:                                    # Binomial(k,1/2) != 0 positions (main.py:371-372, :389)
                    mask = _u32(key, n, 0xFFFF0000 + a) & ((1 << k) - 1)
                    a += 1
                for trial in range(MAX_TRIALS):
                    cand = list(orig)
                    for p_ in range(k):
                        if (mask >> p_) & 1:
                            c = int(node2chrom[orig[p_]])
                            start, end = int(chrom_range[c][0]), int(chrom_range[c][1])
                            r = _u32(key, n, 8 * trial + p_)
                            cand[p_] = start + ((r * (end - start)) >> 32)          # main.py:405-407
                    cand.sort()                                                        # main.py:416
                    gaps = [cand[t + 1] - cand[t] for t in range(k - 1)]
                    if any(g == 0 or g <= min_dis for g in gaps):                      # main.py:410-414, :417-421
                        continue
                    if tuple(cand) in known:                                           # main.py:392
                        continue
                    result = cand
                    break
            out[n, :k] = result
    return out


def assemble_batch(pos: np.ndarray, pos_w: np.ndarray, neg: np.ndarray):
    """x = cat(pos, neg); y = [1..;0..]; w = [pos_w..;1..]  (main.py:443-448, task_mode 'class')."""
    x = np.concatenate([pos, neg], axis=0)
    y = np.concatenate([np.ones((len(pos), 1), np.float32), np.zeros((len(neg), 1), np.float32)])
    w = np.concatenate([np.asarray(pos_w, np.float32).reshape(-1, 1), np.ones((len(neg), 1), np.float32)])
    return x, y, w
