import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matcha_amd import synth, train as T
from tests.test_hip_model import hip_model
num = synth.LAYOUTS["c23"]; N = int(np.sum(num))
rng = np.random.default_rng(3)
edges = np.concatenate([np.pad(synth.make_edges(rng, N, k, 400), ((0, 0), (0, 3 - k))) for k in (2, 3)])
weights = rng.uniform(0.6, 1.0, size=len(edges)).astype(np.float32)
for nb, neps in ((20, 1), (26, 1), (33, 1), (33, 2)):
    res = {}
    for graph in (False, True):
        T.GRAPH_EPOCHS = graph
        np.random.seed(5); torch.manual_seed(5)
        clf, _ = hip_model(num, 64, "adj", 81); clf.train()
        sess = T.Session(clf, synth.node2chrom(num), synth.chrom_range(num).astype(np.int32), 2, 3, 0, seed=11, deterministic=True)
        sess.set_known(edges)
        for _ in range(neps):
            out = T.train_epoch(sess, edges[:nb * 24 + 5], weights[:nb * 24 + 5], 1.0, 0.001, batch_size=24)
        torch.cuda.synchronize()
        res[graph] = (out, {n: p.detach().cpu().clone() for n, p in clf.named_parameters()})
    T.GRAPH_EPOCHS = True
    bad = []
    for n in res[False][1]:
        d = float((res[False][1][n] - res[True][1][n]).abs().max())
        if d > 5e-5:
            bad.append((n.replace("node_embedding.", ""), round(d, 5)))
    print("n_batch", nb, "epochs", neps, "losses", res[False][0][:2], res[True][0][:2], "| params off by > 2e-4:", bad[:6], len(bad))
