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


def _sparse_worker(rank, world, port, out_dir):
    """Row-sparse exchange of the table gradient over gloo: every rank contributes its per-token
    (node id, gradient row) list -- here: the oracle's d loss / d(gathered row) per token of its shard -- padded to a fixed
    capacity with id-0 entries; after parallel.exchange_table_rows every rank holds all lists in rank order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from matcha_amd.parallel import exchange_table_rows
    torch.set_num_threads(2)
    num, d = synth.LAYOUTS["tiny"], 16
    P, fe, _ = oracle_state(num, d, "table", 9, requires_grad=True)
    x, y, w = synth.make_batch(np.random.default_rng(1), int(np.sum(num)), [2, 3, 5], 16)
    idx = shard_rows(len(x), rank, world)
    xs, ys, ws = (torch.from_numpy(a[idx]) for a in (x, y, w))
    # per-token gradient rows: differentiate w.r.t. a per-token copy of the gathered rows
    W = P["node_embedding.weight"]
    tok = xs.reshape(-1)
    rows_in = W.detach()[tok].clone().requires_grad_(True)
    P2 = dict(P)

    class _Fe:                                     # front end that hands out rows_in for this batch
        mode, bounds, n_chrom, n_nodes = "table", fe.bounds, fe.n_chrom, fe.n_nodes
    import oracle.hypersagnn as OH
    orig = OH.node_embeddings
    OH.node_embeddings = lambda P_, fe_, xf, rc=None, am=None: (rows_in * (xf != 0).unsqueeze(-1), torch.zeros(1))
    try:
        loss, *_ = O.total_loss(P2, fe, xs, ys, ws, 1.0, 0.0)
        g_rows, = torch.autograd.grad(loss, [rows_in])
    finally:
        OH.node_embeddings = orig
    cap = xs.numel() + 1                            # B*L + 1, the library's list capacity
    ids = torch.zeros(cap, dtype=torch.int32)
    rows = torch.full((cap, d), float("nan"))       # unused entries hold garbage: they must never be read
    real = tok != 0
    n_real = int(real.sum())
    ids[:n_real] = tok[real].to(torch.int32)
    rows[:n_real] = g_rows[real]
    ids_all, rows_all = exchange_table_rows(ids, rows)
    assert ids_all.shape == (world * cap,) and rows_all.shape == (world * cap, d)
    assert torch.equal(ids_all[rank * cap:(rank + 1) * cap], ids)
    # local reduce (what matcha_scatter_rows does on the device): entries with id 0 are skipped
    dense = torch.zeros(W.shape[0], d, dtype=torch.float64)
    use = ids_all != 0
    dense.index_add_(0, ids_all[use].long(), rows_all[use].double())
    if rank == 0:
        torch.save({"dense": (dense / world).float()}, os.path.join(out_dir, "sparse.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sparse_table_exchange_equals_global_batch_gradient(tmp_path):
    port = _free_port()
    mp.spawn(_sparse_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(tmp_path, "sparse.pt"), weights_only=False)["dense"]
    num, d = synth.LAYOUTS["tiny"], 16
    P, fe, _ = oracle_state(num, d, "table", 9, requires_grad=True)
    x, y, w = synth.make_batch(np.random.default_rng(1), int(np.sum(num)), [2, 3, 5], 16)
    _, _, _, _, grads = O.loss_and_grads(P, fe, torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(w), 1.0, 0.0)
    ref = grads["node_embedding.weight"]
    assert float(ref.abs().max()) > 0
    assert float((got - ref).abs().max()) <= 5e-6 * float(ref.abs().max())
    assert float(got[0].abs().max()) == 0.0


def test_sparse_exchange_decision():
    from matcha_amd.parallel import sparse_exchange_pays
    assert not sparse_exchange_pays(3067, 64, 65536 * 5 + 1, 8)          # hg38 1 Mb: the table is smaller than one rank's list
    assert sparse_exchange_pays(1_000_000, 256, 16384 * 8 + 1, 8)        # BASELINE configs[4]
    assert not sparse_exchange_pays(1_000_000, 256, 16384 * 8 + 1, 1)
