"""Counter-based RNG shared (as a *specification*) by the oracle and the HIP kernels.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference draws dropout masks from torch's global generator
(Modules.py:174, :226-227, :345-346) and negatives from python ``random`` /
numpy's global generator (main.py:371, :389, :407).  None of those streams can
be reproduced on a GPU, so this project *defines* its random streams as a pure
function of (seed, stream id, counter); the HIP kernels
(matcha_amd/csrc/rng.hpp) and this numpy restatement compute the same bits, so
dropout masks and sampled negatives are bit-comparable between the HIP path and
the oracle.  The *distribution* (Bernoulli(1-p) keep mask scaled by 1/(1-p);
the sampler's accept/reject rules) is what is checked against the reference.

    lowbias32(x):  x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16
    key(seed64, stream) = lowbias32(seed_lo ^ lowbias32(seed_hi ^ lowbias32(stream + 0x9E3779B9)))
    rand_u32(key, hi, lo) = lowbias32(lo ^ lowbias32(hi ^ key))
"""
import numpy as np

STREAM_DROP_ADJ = 1   # Modules.py:186  dropout(0.2) on gathered feature rows;  counter = (token slot, column)
STREAM_DROP_FC1 = 2   # Modules.py:572  dropout(0.3) on fc1 output;            counter = (token slot, feature)
STREAM_DROP_PFF = 3   # Modules.py:359-360 dropout(0.4) inside pff_n1;          counter = (token slot, feature)
STREAM_NEG = 16       # main.py:361-459 negative sampler;                       counter = (negative id, draw index)
STREAM_SHUFFLE = 17   # utils.py:142-149 sync_shuffle / DataGenerator.shuffle

_M32 = np.uint64(0xFFFFFFFF)


def lowbias32(x):
    x = np.asarray(x, dtype=np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def make_key(seed, stream):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    k = lowbias32((int(stream) + 0x9E3779B9) & 0xFFFFFFFF)
    k = lowbias32(np.uint64(hi) ^ k)
    k = lowbias32(np.uint64(lo) ^ k)
    return np.uint64(k)


def rand_u32(key, hi, lo):
    hi = np.asarray(hi, dtype=np.uint64) & _M32
    lo = np.asarray(lo, dtype=np.uint64) & _M32
    return lowbias32(lo ^ lowbias32(hi ^ np.uint64(key))).astype(np.uint32)


def dropout_threshold(p):
    """keep  <=>  rand_u32 >= threshold ;  threshold = floor(p * 2^32)."""
    return np.uint32(min(int(float(p) * 4294967296.0), 0xFFFFFFFF))


def dropout_mask(seed, stream, p, n_rows, n_cols, row_ids=None):
    """Multiplier mask [n_rows, n_cols] float32: 0 where dropped, 1/(1-p) where kept.

    ``row_ids`` (default arange(n_rows)) are the token slots used as the counter's
    high word, so a mask for a subset of tokens equals the matching rows of the
    full mask.
    """
    if p <= 0.0:
        return np.ones((n_rows, n_cols), dtype=np.float32)
    key = make_key(seed, stream)
    rows = np.arange(n_rows, dtype=np.uint64) if row_ids is None else np.asarray(row_ids, dtype=np.uint64)
    cols = np.arange(n_cols, dtype=np.uint64)
    r = rand_u32(key, rows[:, None], cols[None, :])
    keep = r >= dropout_threshold(p)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(keep, scale, np.float32(0.0)).astype(np.float32)


def rand_float01(key, hi, lo):
    """Uniform in [0,1): top 24 bits / 2^24 (exact in float32)."""
    return (rand_u32(key, hi, lo) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
