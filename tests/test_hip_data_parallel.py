"""The Trainer's data-parallel step on the GPU with two ranks.  Both ranks share cuda:0 and talk over
gloo (one MI355X on the test box, so RCCL cannot place two ranks; the collective calls are the same torch.distributed ops
the RCCL launch uses): broadcast of rank 0's weights, per-rank shard of the batch, ONE all-reduce bucket of gradients +
touched flags, recon-gradient weighting in adj mode.  The result must equal the single-rank step on the global batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from matcha_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(num):
    x, y, w = synth.make_batch(np.random.default_rng(4), int(np.sum(num)), [2, 3, 5], 40)      # 120 rows
    return x, y.reshape(-1), w.reshape(-1)


def _worker(rank, world, port, out_dir, mode, d):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from matcha_amd.engine import Trainer
    from matcha_amd.parallel import shard_rows
    from tests.test_hip_model import hip_model
    num = synth.LAYOUTS["tiny"]
    clf, _ = hip_model(num, d, mode, 50)                   # same frozen features everywhere (they are data, not parameters)
    clf.eval()                                              # no dropout: masks are indexed by the LOCAL batch slot
    if rank == 1:                                           # different weights on rank 1: the Trainer must broadcast rank 0's
        with torch.no_grad():
            clf._runtime().flat.add_(0.25)
    tr = Trainer(clf, lr=1e-3)
    x, y, w = _batch(num)
    idx = shard_rows(len(x), rank, world)
    xs, ys, ws = (torch.from_numpy(a[idx]).cuda() for a in (x, y, w))
    # step 0 taken apart so that the EXCHANGED gradient can be compared (parameters after AdamW amplify rounding noise near eps)
    tr.forward_backward(xs, ys, ws, 1.0, 0.3, 1)
    tr.all_reduce()
    torch.cuda.synchronize()
    g0 = (tr.gflat / world).cpu()
    t0 = tr.touched.cpu()
    tr.optimizer_step()
    tr.step(xs, ys, ws, alpha=1.0, beta=0.3, random_chrom=1)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"params": {n: p.detach().cpu() for n, p in clf.named_parameters()}, "g0": g0, "touched": t0}, os.path.join(out_dir, "dp.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,d", [("table", 64), ("adj", 64), ("adj", 16)])
def test_two_rank_trainer_equals_single_rank_on_global_batch(tmp_path, mode, d):
    from matcha_amd.engine import Trainer
    from tests.test_hip_model import GAUGE, hip_model
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode, d), nprocs=2, join=True)
    got = torch.load(os.path.join(tmp_path, "dp.pt"), weights_only=False)
    num = synth.LAYOUTS["tiny"]
    clf, _ = hip_model(num, d, mode, 50)
    clf.eval()
    tr = Trainer(clf, lr=1e-3)
    x, y, w = _batch(num)
    xt, yt, wt = (torch.from_numpy(a).cuda() for a in (x, y, w))
    tr.forward_backward(xt, yt, wt, 1.0, 0.3, 1)
    torch.cuda.synchronize()
    g_ref, t_ref = tr.gflat.cpu(), tr.touched.cpu()
    tr.optimizer_step()
    tr.step(xt, yt, wt, alpha=1.0, beta=0.3, random_chrom=1)
    torch.cuda.synchronize()
    # the exchanged gradient (sum over the two ranks / 2) == the single-rank gradient on the global batch, tensor by tensor, and
    # "grad is None" (the touched flags) is decided on the global batch
    assert torch.equal(got["touched"], t_ref)
    rt = tr.rt
    for p_, o in zip(rt.live, rt.seg_off_list[:-1]):
        ref = g_ref[o:o + p_.numel()]
        # (absolute floor: one tensor's true gradient is zero -- the gauge direction, DESIGN.md §2 -- and holds rounding noise ~1e-9)
        assert float((got["g0"][o:o + p_.numel()] - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 2e-8, o
    for n, p in clf.named_parameters():
        if n == GAUGE:
            continue
        a, b = p.detach().cpu(), got["params"][n]
        diff, scale = (a - b).abs().reshape(-1), max(1.0, float(a.abs().max()))
        # the two runs sum the same fp32 terms in a different order; AdamW's g / (|g| + eps) turns that rounding noise into a
        # visible fraction of lr only where |g| is itself near eps: the typical element agrees to 2e-5, none moves further apart
        # than the 2 * lr per step such an element can (the gradient comparison above is the sharp statement)
        assert float(torch.quantile(diff, 0.5)) <= 2e-5 * scale, (n, float(torch.quantile(diff, 0.5)))
        assert float(diff.max()) <= 2 * 1e-3 * 2 + 1e-5 * scale, (n, float(diff.max()))


def _sparse_worker(rank, world, port, out_dir, exchange, d):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from matcha_amd.engine import Trainer
    from matcha_amd.parallel import shard_rows
    from tests.test_hip_model import hip_model
    num = synth.LAYOUTS["c1"]
    clf, _ = hip_model(num, d, "table", 50)
    clf.eval()
    tr = Trainer(clf, lr=1e-3, table_exchange=exchange)
    rng = np.random.default_rng(8)
    N = int(np.sum(num))
    for step in range(6):
        x, y, w = synth.make_batch(rng, N, [2, 3, 4, 5], 64)                   # 256 rows, the same global batch on both ranks
        idx = shard_rows(len(x), rank, world)
        xs, ys, ws = (torch.from_numpy(a[idx]).cuda() for a in (x, y.reshape(-1), w.reshape(-1)))
        if step == 0:
            # the step taken apart: the exchanged gradient (sum over ranks, scaled by 1/world in AdamW) is what must equal the
            # single-rank gradient on the global batch; parameters after AdamW amplify rounding noise wherever |g| ~ eps
            tr.forward_backward(xs, ys, ws, 1.0, 0.0, 0)
            tr.all_reduce()
            torch.cuda.synchronize()
            g0 = (tr.gflat / world).cpu()
            tr.optimizer_step()
        else:
            tr.step(xs, ys, ws, alpha=1.0, beta=0.0)
        assert tr._sparse == (exchange == "sparse")
    torch.cuda.synchronize()
    tr.check_status()
    if rank == 0:
        torch.save({"params": {n: p.detach().cpu() for n, p in clf.named_parameters()}, "g0": g0}, os.path.join(out_dir, f"dp_{exchange}.pt"))
    # both ranks must hold the same parameters bit for bit (same gradient, same AdamW)
    flat = clf._runtime().flat.detach().cpu()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("d", [64, 128])
def test_two_rank_sparse_table_exchange_equals_dense_and_single_rank(tmp_path, d):
    """all-gather of per-token (id, row) lists + deterministic local sum (Trainer table_exchange="sparse")
    against the flat all-reduce ("dense") and against the single-rank run on the global batch, six AdamW steps; two runs of the
    sparse path are bitwise equal (the dense path's table comes out of a sum whose order gloo fixes as well)."""
    from matcha_amd.engine import Trainer
    from tests.test_hip_model import GAUGE, hip_model
    runs = {}
    for tag, exchange in (("sparse", "sparse"), ("sparse2", "sparse"), ("dense", "dense")):
        out = tmp_path / tag
        out.mkdir()
        mp.spawn(_sparse_worker, args=(2, _free_port(), str(out), exchange, d), nprocs=2, join=True)
        runs[tag] = torch.load(os.path.join(out, f"dp_{exchange}.pt"), weights_only=False)
    for n in runs["sparse"]["params"]:
        assert torch.equal(runs["sparse"]["params"][n], runs["sparse2"]["params"][n]), n      # bitwise reproducible, table included
    assert torch.equal(runs["sparse"]["g0"], runs["sparse2"]["g0"])
    num = synth.LAYOUTS["c1"]
    N = int(np.sum(num))
    clf, _ = hip_model(num, d, "table", 50)
    clf.eval()
    tr = Trainer(clf, lr=1e-3)
    rng = np.random.default_rng(8)
    for step in range(6):
        x, y, w = synth.make_batch(rng, N, [2, 3, 4, 5], 64)
        xt, yt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(y.reshape(-1)).cuda(), torch.from_numpy(w.reshape(-1)).cuda()
        if step == 0:
            tr.forward_backward(xt, yt, wt, 1.0, 0.0, 0)
            torch.cuda.synchronize()
            g_ref = tr.gflat.cpu()
            tr.optimizer_step()
        else:
            tr.step(xt, yt, wt, alpha=1.0, beta=0.0)
    torch.cuda.synchronize()
    # (1) the exchanged gradient of step 0 == the single-rank gradient on the global batch, tensor by tensor
    rt = tr.rt
    for p_, o in zip(rt.live, rt.seg_off_list[:-1]):
        ref = g_ref[o:o + p_.numel()]
        for tag in ("sparse", "dense"):
            got = runs[tag]["g0"][o:o + p_.numel()]
            assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 2e-8, (tag, o)
    # (2) parameters after six AdamW steps: the typical element agrees closely; where |g| is within a few
    #     orders of eps the rounding noise of the two summation orders moves an element by up to lr per step in either direction
    for n, p in clf.named_parameters():
        if n == GAUGE:
            continue
        a = p.detach().cpu()
        for tag in ("sparse", "dense"):
            b = runs[tag]["params"][n]
            diff, scale = (a - b).abs().reshape(-1), max(1.0, float(a.abs().max()))
            assert float(torch.quantile(diff[:1 << 20], 0.5)) <= 2e-5 * scale, (tag, n)      # the typical element; the tails are Adam noise
            assert float(diff.max()) <= 2 * 1e-3 * 6 + 1e-5 * scale, (tag, n, float(diff.max()))


def _nccl_worker(rank, world, port, out_dir):
    """ONE rank over the real RCCL backend (the test box has one GPU): every collective of both exchange forms runs through
    torch.distributed's nccl backend on device buffers (all-reduce of the encoder tail on the side stream behind the
    encoder_done event, all-reduce of the front part, all-gather of the 4-byte counts and of the compacted (id, row) lists) and,
    with one rank, must be the identity: parameters bitwise equal to the step without collectives."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from matcha_amd.engine import Trainer
    from tests.test_hip_model import hip_model
    num = synth.LAYOUTS["c1"]
    N = int(np.sum(num))
    report = {}
    for d, exchange, overlap, compact in ((64, "dense", True, True), (64, "dense", False, True), (64, "sparse", True, True),
                                          (128, "sparse", True, True), (128, "sparse", False, False), (128, "dense", True, True),
                                          (128, "sparse", True, "host")):     # "host": the caller passes the token count (no device read-back)
        flats = []
        host_count = compact == "host"
        compact = bool(compact)
        for forced in (False, True):
            clf, _ = hip_model(num, d, "table", 50)
            clf.train()                                              # dropout on: the same (seed, slot) masks in both runs
            tr = Trainer(clf, lr=1e-3, base_seed=5, table_exchange=exchange, deterministic=True)
            tr.force_collectives = forced
            tr.overlap_exchange, tr.compact_exchange = overlap, compact
            rng = np.random.default_rng(8)
            for step in range(4):
                x, y, w = synth.make_batch(rng, N, [2, 3, 4, 5], 64)
                tr.step(torch.from_numpy(x).cuda(), torch.from_numpy(y.reshape(-1)).cuda(), torch.from_numpy(w.reshape(-1)).cuda(), alpha=1.0, beta=0.0,
                        max_tokens=int((x != 0).sum()) if host_count else None)
            torch.cuda.synchronize()
            tr.check_status()
            flats.append(clf._runtime().flat.detach().cpu())
            if forced:
                assert tr._sparse == (exchange == "sparse")
                cb = dict(tr.comm_bytes)
                if exchange == "sparse":
                    assert cb["table_rows_allgather"] == 0 and (cb["table_rows_fill"] < 0.9) == compact      # (world - 1) = 0 peers; mixed k: ~70 % fill
                assert ("encoder_allreduce" in cb) == overlap and ("bucket_allreduce" in cb) == (not overlap)
                report[f"{d}-{exchange}-{overlap}-{compact}-{host_count}"] = cb
        assert torch.equal(flats[0], flats[1]), (d, exchange, overlap, compact)
    import json
    json.dump(report, open(os.path.join(out_dir, "nccl.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_one_rank_both_exchange_forms_are_identity(tmp_path):
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(os.path.join(tmp_path, "nccl.json"))


def test_bench_refuses_to_report_more_gpus_than_it_ran(tmp_path):
    """`python bench.py --gpus N` without a launcher starts the N ranks itself (one child torch.distributed.run before this process
    touches the GPU); on a box with fewer GPUs it must exit non-zero and print NO record -- never an n_gpus: 1 line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], capture_output=True,
                         text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert run.returncode != 0
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert "refusing" in run.stderr


def test_bench_one_rank_launch_goes_through_rccl(tmp_path):
    """The driver's N > 1 command line with N = 1 and the default backend: bench.py joins an nccl (= RCCL) process group and every
    step runs its collectives; the record says which and how many bytes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("MATCHA_DIST_BACKEND", "MATCHA_LOCAL_DEVICE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for extra, form in (([], "flat all-reduce"), (["--table-exchange", "sparse"], "row-sparse all-gather")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
               "--rows", "8192", "--edges-per-k", "20000", "--no-extras", "--no-cpu-baseline"] + extra
        run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
        assert run.returncode == 0, run.stderr[-2000:]
        rec = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
        assert rec["n_gpus"] == 1 and rec["config"]["table_gradient_exchange"] == form
        cb = rec["config"]["collective_payload_bytes_per_step"]
        assert cb["encoder_allreduce"] > 0 and rec["config"]["exchange_overlapped"] is True
        assert ("table_rows_allgather" in cb) == (form == "row-sparse all-gather")


def _run_worker(rank, world, port, tmp, front_end):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MATCHA_DIST_BACKEND="gloo", MATCHA_LOCAL_DEVICE="0")
    import json
    from matcha_amd import train as T
    r, w_, device = T.init_distributed()
    assert (r, w_) == (rank, world)
    np.random.seed(100 + rank)                                   # ranks start from DIFFERENT host RNG states: run() must align them
    torch.manual_seed(200 + rank)
    cfg = json.load(open(os.path.join(tmp, "config.JSON")))
    logs = []
    model = T.run(cfg, front_end=front_end, epochs1=1, epochs2=2, batches_per_epoch=3, device=device,
                  emb_path=os.path.join(tmp, "embeddings.npy"), log=logs.append)
    flat = model._runtime().flat.detach().cpu()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])                         # replicas stayed in lockstep through both phases
    # numpy's global stream (epoch permutations, DataGenerator shuffles, random_chrom) is still shared after both phases: root-only
    # work that draws from it (save_embeddings with the adj front end) would have pushed rank 0 ahead
    nxt = torch.from_numpy(np.random.randint(0, 1 << 30, size=8))
    seen = [torch.zeros_like(nxt) for _ in range(world)]
    dist.all_gather(seen, nxt)
    assert torch.equal(seen[0], seen[1]), "the ranks' numpy streams drifted apart"
    # the epoch loop replayed captured steps on every rank (graph A: sampling + forward + backward; the exchange; graph B: AdamW + records) --
    # except where the step has a collective of its own in the middle (adj front end, beta != 0), which runs call by call
    json.dump({"train_graph_replays": T.STATS["train_graph_replays"]}, open(os.path.join(tmp, f"graphs{rank}.json"), "w"))
    if rank == 0:
        assert any("Training" in l for l in logs)
        json.dump({"n_logs": len(logs)}, open(os.path.join(tmp, "rank0.json"), "w"))
    else:
        assert not logs                                          # only rank 0 reports
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("front_end", ["table", "adj"])
def test_train_run_data_parallel_two_ranks(tmp_path, front_end):
    """matcha_amd.train.run (main.py:516-685's flow) under a two-rank launch: shared numpy stream (split, shuffles, random_chrom),
    strided shards of each global batch, rank-0-only file writes, replicas bitwise identical at the end."""
    from tests.test_train_driver import _write_temp_dir
    cfg, num = _write_temp_dir(str(tmp_path), m=600)
    mp.spawn(_run_worker, args=(2, _free_port(), str(tmp_path), front_end), nprocs=2, join=True)
    N = int(np.sum(num))
    emb = np.load(os.path.join(tmp_path, "embeddings.npy"))
    assert emb.shape == (N, 16) and np.isfinite(emb).all()
    assert os.path.exists(os.path.join(cfg["temp_dir"], "model.chkpt")) and os.path.exists(os.path.join(cfg["temp_dir"], "model2load"))
    assert os.path.exists(os.path.join(tmp_path, "rank0.json"))
    import json
    replays = [json.load(open(os.path.join(tmp_path, f"graphs{r}.json")))["train_graph_replays"] for r in (0, 1)]
    assert replays[0] == replays[1]
    if front_end == "table":
        assert replays[0] >= 2 * (1 + 2) * 1                      # three epochs of three steps: two call by call (warm-up + capture), the rest replayed in pairs
    else:
        assert replays[0] == 0


def test_bench_multi_rank_path_two_ranks_on_one_gpu(tmp_path):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one JSON line from rank 0), here with two
    ranks sharing device 0 over gloo: sharding of the positives, the barrier-bracketed timed region, the max over ranks, the
    per-step gradient all-reduce and rank 0's record."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MATCHA_DIST_BACKEND="gloo", MATCHA_LOCAL_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--rows", "8192", "--edges-per-k", "20000"]
    run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-2000:]
    line = [l for l in run.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["config"]["global_rows_per_step"] == 2 * rec["config"]["rows_per_gpu_per_step"] == 16384
    assert rec["config"]["parallelism"] == "dp2" and rec["config"]["table_gradient_exchange"] == "flat all-reduce"
    assert abs(rec["value"] - 16384 * 3 / (rec["ms_per_step"] * 3e-3)) <= 1e-3 * rec["value"]
    assert "cpu_baseline" not in rec or rec["cpu_baseline"] is None          # rank 0 at N = 1 only
    assert rec["roofline"] is not None and 0.0 < rec["last_bce"] < 5.0
    # the line diagnoses a multi-rank run by itself (the first real 8-GPU run must not need a second one to be understood): every rank
    # reports the process group it saw, the device it ran on and its per-collective time
    coll = rec["config"]["collectives"]
    assert coll["world_size_seen_by_rank"] == [2, 2] and coll["device_by_rank"] == [0, 0] and coll["backend"] == "gloo"
    per_call = coll["collective_us_per_call_by_rank"]
    assert per_call and all(len(v) == 2 and all(t > 0.0 for t in v) for v in per_call.values()), per_call
    assert rec["config"]["collective_payload_bytes_per_step"] and all(b > 0 for b in rec["config"]["collective_payload_bytes_per_step"].values())
