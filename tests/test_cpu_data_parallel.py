"""N>1 path on CPU: two gloo ranks, each with its shard of the batch, must reproduce the one-rank step on the global
batch.  The per-rank compute here is the oracle (this is a test); what is under test is the
product's sharding + gradient exchange logic (matcha_amd/parallel.py), which the Trainer uses unchanged with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from matcha_amd import synth
from matcha_amd.parallel import allreduce_bucket, allreduce_gradients, broadcast_parameters, recon_grad_weight, shard_edges, shard_rows
from oracle import hypersagnn as O
from tests.helpers import oracle_state


BETA = 0.37       # weight of the reconstruction loss in the adj-mode check (any non-zero value)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flatten(grads, names, P):
    """fixed layout, zeros where a tensor got no gradient -- like the Trainer's flat gradient buffer"""
    return torch.cat([(grads[n] if grads[n] is not None else torch.zeros_like(P[n])).reshape(-1) for n in names])


def _worker(rank, world, port, out_dir, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    num, d = synth.LAYOUTS["tiny"], 16
    P, fe, _ = oracle_state(num, d, mode, 9, requires_grad=True)
    # rank 1 starts from perturbed weights; broadcast must make them rank 0's
    names = [n for n, t in P.items() if t.requires_grad]
    flat = torch.cat([P[n].detach().reshape(-1) for n in names]).clone()
    if rank == 1:
        flat += 0.5
    broadcast_parameters(flat, 0)
    ref_flat = torch.cat([P[n].detach().reshape(-1) for n in names])
    assert torch.equal(flat, ref_flat)
    x, y, w = synth.make_batch(np.random.default_rng(1), int(np.sum(num)), [2, 3, 5], 16)     # global batch: 48 rows
    idx = shard_rows(len(x), rank, world)
    xs, ys, ws = (torch.from_numpy(a[idx]) for a in (x, y, w))
    beta = 0.0
    if mode == "adj":
        # the recon loss is a mean over this rank's m "other" tokens (real, outside the drawn chromosome 1): weight it so
        # that the averaged gradient is the one of the global mean (what matcha_forward reports in losses[2])
        n2c = synth.node2chrom(num)[xs.numpy()]
        m_local = torch.tensor([float(((n2c >= 0) & (n2c != 1)).sum())])
        beta = float(recon_grad_weight(m_local, BETA)[0])
    loss, bce, recon, logits, grads = O.loss_and_grads(P, fe, xs, ys, ws, 1.0, beta, random_chrom=1)
    gflat = _flatten(grads, names, P)
    touched = torch.tensor([1 if grads[n] is not None else 0 for n in names], dtype=torch.int32)
    # the Trainer's path: one bucket [gradients | touched flags as floats], one SUM all-reduce
    gbuf = torch.cat([gflat, torch.zeros(len(names))])
    t2 = touched.clone()
    assert allreduce_bucket(gbuf, gflat.numel(), t2) == 0.5
    scale = allreduce_gradients(gflat, touched)          # the two-collective form (SUM + MAX) must agree with it
    assert scale == 0.5
    assert torch.equal(gbuf[:gflat.numel()], gflat) and torch.equal(t2, touched)
    if rank == 0:
        torch.save({"g": gflat * scale, "names": names, "touched": touched}, os.path.join(out_dir, "dp.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["table", "adj"])
def test_two_rank_gradient_equals_global_batch(tmp_path, mode):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    got = torch.load(os.path.join(tmp_path, "dp.pt"), weights_only=False)
    num, d = synth.LAYOUTS["tiny"], 16
    P, fe, _ = oracle_state(num, d, mode, 9, requires_grad=True)
    x, y, w = synth.make_batch(np.random.default_rng(1), int(np.sum(num)), [2, 3, 5], 16)
    loss, bce, recon, logits, grads = O.loss_and_grads(P, fe, torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(w), 1.0,
                                                       BETA if mode == "adj" else 0.0, random_chrom=1)
    names = got["names"]
    ref = _flatten(grads, names, P)
    assert (got["g"] - ref).abs().max() <= 5e-6 * max(1.0, float(ref.abs().max()))
    # grad-None-ness is decided on the GLOBAL batch: touched (MAX over ranks) == "not None" of the one-rank run
    assert got["touched"].tolist() == [1 if grads[n] is not None else 0 for n in names]


def test_shards_are_equal_and_disjoint():
    for n, world in ((1000, 8), (1001, 4), (7, 2)):
        seen = []
        for r in range(world):
            idx = shard_rows(n, r, world)
            assert len(idx) == n // world
            seen.append(idx)
        allidx = np.concatenate(seen)
        assert len(np.unique(allidx)) == len(allidx) and allidx.max() < n
    e = np.arange(40).reshape(20, 2)
    ee, ww = shard_edges(e, np.arange(20, dtype=np.float32), 1, 4)
    assert np.array_equal(ee[:, 0] // 2, ww.astype(int)) and len(ee) == 5
