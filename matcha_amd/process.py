"""The reference's ``process.py`` up to the files the training flow reads: node dictionaries, the cluster
file -> ``edge_list.npy``, cooler pixels -> ``intra_adj.npy`` / ``inter_adj.npy``.

  build_node_dict   process.py:10-39    chrom sizes -> bin2node / node2bin / node2chrom / chrom_range (node ids from 1;
                                        ceil(size / res) + 1 bins per chromosome, as the reference's ``range(max_bin + 1)``)
  parse_clusters    process.py:42-87    SPRITE-style cluster file (id <tab> chr:pos <tab> ...) -> sorted unique node lists
  cool_index2node   process.py:121-137  cooler bin table -> node id per cooler bin (0 = not in chrom_list)
  pixels_to_adj     process.py:144-176  pixels -> intra / inter adjacency, accumulated on the device (csrc/features.hip)

Reading the .mcool container itself (h5py) is out of scope: ``pixels_to_adj`` takes the arrays ``f['pixels']['bin1_id']``,
``['bin2_id']`` and ``['balanced']`` / ``['count']`` hold, in chunks of any size.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib


def build_node_dict(chrom_size_path: str, chrom_list: Sequence[str], res: int, temp_dir: Optional[str] = None):
    """process.py:10-39.  The size file is tab-separated ``chrom <tab> size`` without a header; a chromosome listed twice
    takes its largest size (:22).  Returns (bin2node, node2bin, node2chrom, chrom_range int64 [C, 2]) and, with ``temp_dir``,
    writes the four .npy files the rest of the flow loads."""
    sizes: Dict[str, int] = {}
    with open(chrom_size_path) as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) < 2 or not parts[0]:
                continue
            sizes[parts[0]] = max(sizes.get(parts[0], 0), int(parts[1]))
    bin2node, node2bin, node2chrom, chrom_range = {}, {}, {}, []
    count = 1
    for j, chrom in enumerate(chrom_list):
        if chrom not in sizes:
            raise ValueError("%s is not in %s" % (chrom, chrom_size_path))      # the reference: np.max of an empty selection
        max_bin = math.ceil(sizes[chrom] / res)
        start = count
        for i in range(max_bin + 1):
            b = "%s:%d" % (chrom, i * res)
            bin2node[b] = count
            node2bin[count] = b
            node2chrom[count] = j
            count += 1
        chrom_range.append([start, count])
    chrom_range = np.asarray(chrom_range)
    if temp_dir is not None:
        os.makedirs(temp_dir, exist_ok=True)
        np.save(os.path.join(temp_dir, "chrom_range.npy"), chrom_range)
        np.save(os.path.join(temp_dir, "bin2node.npy"), bin2node)
        np.save(os.path.join(temp_dir, "node2chrom.npy"), node2chrom)
        np.save(os.path.join(temp_dir, "node2bin.npy"), node2bin)
    return bin2node, node2bin, node2chrom, chrom_range


def parse_clusters(cluster_path: str, bin2node: Dict[str, int], chrom_list: Sequence[str], res: int, max_cluster_size: int,
                   temp_dir: Optional[str] = None) -> List[List[int]]:
    """process.py:42-87: the first column is the cluster id; lines with fewer than 2 or more than 50 * max_cluster_size items
    are skipped before parsing, items of other chromosomes are dropped, positions are floored to their bin, node ids are
    de-duplicated; clusters with more than max_cluster_size or fewer than 2 nodes are dropped; sorted ascending."""
    final = []
    chrom_set = set(chrom_list)
    with open(cluster_path, "r") as f:
        for line in f:
            info_list = line.strip().split("\t")[1:]
            if len(info_list) < 2 or len(info_list) > max_cluster_size * 50:
                continue
            temp = []
            for info in info_list:
                try:
                    chrom, bin_ = info.split(":")
                except ValueError:
                    raise EOFError(info)
                if chrom not in chrom_set:
                    continue
                b = int(math.floor(int(bin_) / res)) * res
                temp.append(bin2node["%s:%d" % (chrom, b)])
            temp = sorted(set(temp))
            if len(temp) > max_cluster_size:
                continue
            if len(temp) > 1:
                final.append(temp)
    if temp_dir is not None:
        arr = np.empty(len(final), dtype=object)
        for i, c in enumerate(final):
            arr[i] = c
        np.save(os.path.join(temp_dir, "edge_list.npy"), arr, allow_pickle=True)
    return final


def cool_index2node(bins_chrom: np.ndarray, bins_start: np.ndarray, chrom_names: Sequence[str], chrom_list: Sequence[str],
                    bin2node: Dict[str, int]) -> np.ndarray:
    """process.py:121-137 as an int32 array: node id of every cooler bin, 0 for bins of chromosomes outside chrom_list.
    A bin of a listed chromosome that the node dictionary lacks raises KeyError, as the reference's dict lookup does."""
    names = [n.decode() if isinstance(n, bytes) else str(n) for n in chrom_names]
    listed = set(chrom_list)
    out = np.zeros(len(bins_chrom), dtype=np.int32)
    for i in range(len(bins_chrom)):
        chrom = names[int(bins_chrom[i])]
        if chrom in listed:
            out[i] = bin2node["%s:%d" % (chrom, int(bins_start[i]))]
    return out


def node2chrom_array(node2chrom: Dict[int, int], n_nodes: int) -> np.ndarray:
    out = np.full(n_nodes + 1, -1, dtype=np.int32)
    for k, v in node2chrom.items():
        if 0 < int(k) <= n_nodes:
            out[int(k)] = int(v)
    return out


def pixels_to_adj(bin1, bin2, count, index2node, node2chrom, n_nodes: int, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                  device="cuda") -> Tuple[torch.Tensor, torch.Tensor]:
    """process.py:144-172 on the device.  ``n_nodes`` = number of nodes N (the reference's matrices are
    [max(chrom_range) - 1]^2 = N x N, row = node id - 1).  Adds into ``out`` = (intra, inter) float64 [N, N] when given, so
    a large pixel table can be streamed through in chunks.  Each (bin1, bin2) of a cooler occurs once, so every cell receives
    one add (two on the diagonal) and the result does not depend on the order of the atomics."""
    lib = _lib.load()
    dev = torch.device(device)
    def dev_array(a, dtype):                               # numpy arrays, h5py datasets or tensors already on the device
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
        return t.to(device=dev, dtype=dtype).contiguous()
    b1, b2 = dev_array(bin1, torch.int64), dev_array(bin2, torch.int64)
    cnt, i2n = dev_array(count, torch.float64), dev_array(index2node, torch.int32)
    n2c = node2chrom_array(node2chrom, n_nodes) if isinstance(node2chrom, dict) else np.asarray(node2chrom, dtype=np.int32)
    n2c = torch.from_numpy(n2c).to(dev)
    if len(n2c) < n_nodes + 1:
        raise ValueError("node2chrom must cover node ids 0 .. n_nodes")
    if out is None:
        out = (torch.zeros((n_nodes, n_nodes), dtype=torch.float64, device=dev), torch.zeros((n_nodes, n_nodes), dtype=torch.float64, device=dev))
    intra, inter = out
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.matcha_pixels_to_adj(_lib.ptr(b1), _lib.ptr(b2), _lib.ptr(cnt), b1.numel(), _lib.ptr(i2n), i2n.numel(), _lib.ptr(n2c),
                                        n_nodes, _lib.ptr(intra), _lib.ptr(inter), st), "matcha_pixels_to_adj")
    return intra, inter


def save_adj(temp_dir: str, intra: torch.Tensor, inter: torch.Tensor):
    """process.py:175-176."""
    np.save(os.path.join(temp_dir, "intra_adj.npy"), intra.cpu().numpy())
    np.save(os.path.join(temp_dir, "inter_adj.npy"), inter.cpu().numpy())


def main(argv=None):
    """The script body of process.py:229-242: node dictionaries, edge_list.npy and -- when h5py is importable, it is not part
    of this image -- the adjacency matrices from ``mcool_path``, streamed through the device 16 M pixels at a time."""
    import argparse
    import json
    ap = argparse.ArgumentParser(description="process.py on the MI355X path")
    ap.add_argument("--config", default="./config.JSON")
    args = ap.parse_args(argv)
    with open(args.config) as f:
        config = json.load(f)
    res, chrom_list, temp_dir = config["resolution"], config["chrom_list"], config["temp_dir"]
    bin2node, _, node2chrom, chrom_range = build_node_dict(config["chrom_size"], chrom_list, res, temp_dir)
    clusters = parse_clusters(config["cluster_path"], bin2node, chrom_list, res, config["max_cluster_size"], temp_dir)
    print("%d nodes, %d clusters" % (int(chrom_range.max()) - 1, len(clusters)))
    try:
        import h5py
    except ImportError:
        raise SystemExit("h5py is not installed: load the cooler's bins / pixels yourself and call process.pixels_to_adj")
    f = h5py.File(config["mcool_path"], "r")["resolutions"][str(res)]
    i2n = cool_index2node(np.array(f["bins"]["chrom"]), np.array(f["bins"]["start"]), np.array(f["chroms"]["name"]).astype("str"),
                          chrom_list, bin2node)
    weights = f["pixels"]["balanced"] if "balanced" in f["pixels"].keys() else f["pixels"]["count"]       # process.py:147-150
    N, out, step = int(chrom_range.max()) - 1, None, 1 << 24
    for s in range(0, len(weights), step):
        out = pixels_to_adj(f["pixels"]["bin1_id"][s:s + step], f["pixels"]["bin2_id"][s:s + step], weights[s:s + step], i2n, node2chrom, N, out=out)
    save_adj(temp_dir, *out)


if __name__ == "__main__":
    main()
