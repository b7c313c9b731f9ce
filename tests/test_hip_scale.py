"""BASELINE.json configs[3] and [4] at their full sizes on one GPU, through properties that need no oracle run
(the oracle cannot finish these sizes in seconds): 64-bit offsets into the 189 MB per-chromosome feature blocks and the
3.7 GB inter matrix (hg38 100 kb, d = 128, adj), and a 1 M-node, d = 256 table with k up to 8 (1 GB of embeddings + 2 GB of
AdamW moments).  Synthetic inputs are generated on the device."""
import numpy as np
import pytest
import torch

from matcha_amd import synth

pytestmark = pytest.mark.gpu


def _batch(n_nodes, ks, rows_per_k, L, gen):
    """Rows of k distinct ascending node ids, zero-padded to L, shuffled (device)."""
    xs = []
    for k in ks:
        r = torch.randint(1, n_nodes + 1, (rows_per_k * 2, k), device="cuda", generator=gen)
        r = torch.sort(r, dim=1).values
        ok = (r[:, 1:] != r[:, :-1]).all(dim=1)
        r = r[ok][:rows_per_k]
        assert len(r) == rows_per_k
        xs.append(torch.nn.functional.pad(r, (0, L - k)))
    x = torch.cat(xs)
    return x[torch.randperm(len(x), device="cuda", generator=gen)].contiguous()


def _labels(B, gen):
    y = (torch.rand(B, device="cuda", generator=gen) < 0.25).float()
    return y, torch.where(y > 0, torch.full_like(y, 3.0), torch.ones_like(y))


def test_one_million_nodes_d256_table_training_step():
    import Modules as M
    from matcha_amd.engine import Trainer
    from matcha_amd.train import get_attributes
    num = [40000] * 25
    N, d = 1_000_000, 256
    gen = torch.Generator(device="cuda").manual_seed(5)
    torch.manual_seed(5)
    clf = M.Classifier(n_head=8, d_model=d, d_k=d, d_v=d, node_embedding=M.Wrap_Embedding(N + 1, d, padding_idx=0), diag_mask=True,
                       bottle_neck=d, attribute_dict=get_attributes(num)).to("cuda")
    x = _batch(N, [2, 3, 4, 5, 6, 7, 8], 4096, 8, gen)
    y, w = _labels(len(x), gen)
    # inference at this size: deterministic, rows independent at equal width
    clf.eval()
    with torch.no_grad():
        base = clf(x)
        assert base.shape == (len(x), 1) and bool(torch.isfinite(base).all())
        assert torch.equal(base, clf(x))
        assert torch.allclose(clf(x[5000:6000].contiguous()), base[5000:6000], rtol=0, atol=2e-6)
        emb = clf.get_node_embeddings(torch.tensor([[N, 1, N // 2]], device="cuda"))
        assert emb.shape == (1, 3, d) and bool(torch.isfinite(emb).all())
    # ten training steps (Adam's first sign-like steps overshoot at d = 256, then the loss comes down): dense AdamW on the whole table like torch.optim.AdamW on nn.Embedding's dense gradient
    clf.train()
    tr = Trainer(clf, lr=1e-3, weight_decay=1e-2)
    table = clf.node_embedding.weight
    before = table.detach().clone()
    losses = []
    for _ in range(10):
        bce, recon, _ = tr.step(x, y, w, alpha=1.0, beta=0.001)
        losses.append(float(bce))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    after = table.detach()
    seen = torch.zeros(N + 1, dtype=torch.bool, device="cuda")
    seen[x.reshape(-1)] = True
    seen[0] = False
    assert float(after[0].abs().max()) == 0.0                                       # the padding row stays zero
    # rows that never received a gradient only decay: p (1 - lr wd)^10
    untouched = ~seen
    untouched[0] = False
    assert torch.allclose(after[untouched], before[untouched] * (1 - 1e-3 * 1e-2) ** 10, rtol=2e-6, atol=1e-9)
    # rows in the batch moved by at most about lr per step
    delta = (after[seen] - before[seen]).abs()
    assert float(delta.max()) <= 10 * 1.1e-3 + 1e-4 and float(delta.mean()) > 2e-4
    # the highest node id is addressable in gather and scatter
    top = torch.tensor([[N - 1, N], [N - 2, N]], device="cuda")
    b2 = table.detach().clone()
    tr.step(top, torch.ones(2, device="cuda"), torch.ones(2, device="cuda"))
    assert float((table.detach()[N] - b2[N]).abs().max()) > 1e-4


def test_hg38_100kb_d128_adj_training_step():
    import Modules as M
    from matcha_amd.engine import Trainer
    from matcha_amd.train import get_attributes
    num = synth.LAYOUTS["hg38_100kb"]
    N, d = int(np.sum(num)), 128
    assert N > 30000 and max(num) > 2400
    gen = torch.Generator(device="cuda").manual_seed(9)
    torch.manual_seed(9)
    # correlation-like features and a sparse positive inter matrix, generated on the device (189 MB + 3.7 GB)
    feats = [torch.rand((n, n), device="cuda", generator=gen) * 2 - 1 for n in num]
    inter = torch.rand((N, N), device="cuda", generator=gen) * (torch.rand((N, N), device="cuda", generator=gen) < 0.3)
    ne = M.MultipleEmbedding(feats, d, False, torch.as_tensor(np.cumsum(num)), synth.chrom_range(num), inter)
    z = ne.inter_initial.embedding
    assert z.data_ptr() == inter.data_ptr()                                         # z-scored in place on the device
    row = z[N - 1][z[N - 1] != 0]
    assert abs(float(row.mean())) < 1e-4 and abs(float(row.std(unbiased=False)) - 1) < 1e-3
    clf = M.Classifier(n_head=8, d_model=d, d_k=d, d_v=d, node_embedding=ne, diag_mask=True, bottle_neck=d,
                       attribute_dict=get_attributes(num)).to("cuda")
    x = _batch(N, [2, 3, 4, 5], 4096, 5, gen)
    x[0] = torch.tensor([N - 3, N - 2, N - 1, N, 0], device="cuda")                  # last rows of the last feature block
    y, w = _labels(len(x), gen)
    clf.eval()
    with torch.no_grad():
        base = clf(x)
        assert bool(torch.isfinite(base).all()) and torch.equal(base, clf(x))
        assert torch.allclose(clf(x[:777].contiguous()), base[:777], rtol=0, atol=2e-5)
    clf.train()
    tr = Trainer(clf)
    names = [n for n, p in clf.named_parameters() if p.requires_grad]
    before = {n: p.detach().clone() for n, p in clf.named_parameters() if p.requires_grad}
    last = len(num) - 1
    out = []
    for step in range(10):
        bce, recon, _ = tr.step(x, y, w, alpha=1.0, beta=0.001, random_chrom=last)  # the reconstruction head of the last chromosome
        out.append((float(bce), float(recon)))
    assert all(np.isfinite(v) for pair in out for v in pair) and out[-1][0] < out[0][0]
    after = dict(clf.named_parameters())
    for n in names:
        assert bool(torch.isfinite(after[n]).all()), n
    moved = [n for n in names if float((after[n].detach() - before[n]).abs().max()) > 0]
    # every chromosome's encoder saw rows; only the chosen reconstruction head did
    assert all(f"node_embedding.Embedding_Linear{i}.tied weight_0" in moved for i in range(len(num)))
    assert f"node_embedding.Embedding_recon{last}.FF_Linear0.weight" in moved
    assert f"node_embedding.Embedding_recon0.FF_Linear0.weight" not in moved
