"""CPU restatement of the reference's k-mer generation -- TEST INFRASTRUCTURE ONLY.

Reference: Code/generate_kmers.py.  For a k-mer size ``size`` the script keeps the clusters with size <= len <= max_size
(:86-90), and for every node i and every kept cluster containing i (:92-96) forms all ``size - 1``-combinations of the
cluster's nodes greater than ``i + min_dis`` (:17), drops those with a consecutive gap <= min_dis (:24-32, only when
size > 2), counts how often each (i, combination) occurs over all clusters (:34-37) and keeps the ones seen at least
``min_freq_cutoff`` times (:40).  Rows are [i, combination...] (:46-47).  Since clusters are sorted unique node lists
(process.py:66-77) this is: the multiset of ascending k-subsets of each cluster whose adjacent gaps all exceed min_dis,
counted over clusters, thresholded.  The reference's row ORDER depends on worker scheduling (:108-129) and is not part of
the result; this restatement returns rows sorted lexicographically.
"""
from collections import Counter
from itertools import combinations
from typing import Sequence, Tuple

import numpy as np


def generate_kmers(clusters: Sequence[Sequence[int]], size: int, min_dis: int, max_size: int, min_freq_cutoff: int) -> Tuple[np.ndarray, np.ndarray]:
    counter = Counter()
    for datum in clusters:
        datum = np.asarray(datum)
        if not (size <= len(datum) <= max_size):                       # generate_kmers.py:88
            continue
        for i in datum:                                                # node2usefulindex: clusters containing i (:92-96)
            rest = datum[datum > i + min_dis]                          # :17
            for comb in combinations(rest, size - 1):
                if size > 2 and min(comb[j + 1] - comb[j] for j in range(size - 2)) <= min_dis:   # :24-32
                    continue
                counter[(int(i),) + tuple(int(c) for c in comb)] += 1  # :34-36
    keys = sorted(k for k, c in counter.items() if c >= min_freq_cutoff)                         # :40
    if not keys:
        return np.zeros((0, size), dtype=np.int64), np.zeros((0,), dtype=np.int64)
    return np.asarray(keys, dtype=np.int64), np.asarray([counter[k] for k in keys], dtype=np.int64)
