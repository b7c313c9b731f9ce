"""CPU restatement of the reference's contact-map preprocessing -- TEST INFRASTRUCTURE ONLY.

Reference: Code/process.py:144-172 (``parse_cool_contact``'s pixel loop).  The per-chromosome correlation features
(main.py:571-575) and the row z-score of the inter matrix (Modules.py:146-152) are restated in oracle/hypersagnn.py
(``corrcoef_features``, ``zscore_inter``).  Pinned by tests/golden/g9_process.npz, which the reference's own functions
produced (tests/golden/make_golden.py::g9_process).
"""
from typing import Dict, Tuple

import numpy as np


def pixels_to_adj(bin1: np.ndarray, bin2: np.ndarray, count: np.ndarray, index2node: np.ndarray, node2chrom: Dict[int, int],
                  n_nodes: int) -> Tuple[np.ndarray, np.ndarray]:
    """``index2node[i]`` = node id of cooler bin i, 0 when the bin's chromosome is not listed (the reference's dict simply
    lacks those keys, process.py:129-137).  float64 [N, N] matrices, both triangles filled, NaN counts skipped."""
    intra = np.zeros((n_nodes, n_nodes))
    inter = np.zeros((n_nodes, n_nodes))
    for i in range(len(bin1)):                                           # process.py:154
        n1, n2 = int(index2node[bin1[i]]), int(index2node[bin2[i]])
        if n1 < 1 or n2 < 1:                                            # :157-158
            continue
        c = float(count[i])
        if np.isnan(c):                                                  # :165
            continue
        m = intra if node2chrom[n1] == node2chrom[n2] else inter       # :166-173
        m[n1 - 1, n2 - 1] += c
        m[n2 - 1, n1 - 1] += c
    return intra, inter
