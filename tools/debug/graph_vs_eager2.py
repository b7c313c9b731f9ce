import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from matcha_amd import synth, train as T
from tests.test_hip_model import hip_model
num = synth.LAYOUTS["c23"]; N = int(np.sum(num))
rng = np.random.default_rng(3)
edges = np.concatenate([np.pad(synth.make_edges(rng, N, k, 400), ((0, 0), (0, 3 - k))) for k in (2, 3)])
weights = rng.uniform(0.6, 1.0, size=len(edges)).astype(np.float32)
use = edges if len(sys.argv) < 2 else edges[:int(sys.argv[1])]
usew = weights[:len(use)]
res = {}
for tag, graph in (("e1", False), ("e2", False), ("g1", True), ("g2", True)):
    T.GRAPH_EPOCHS = graph
    np.random.seed(5); torch.manual_seed(5)
    clf, _ = hip_model(num, 64, "adj", 81); clf.train()
    sess = T.Session(clf, synth.node2chrom(num), synth.chrom_range(num).astype(np.int32), 2, 3, 0, seed=11, deterministic=True)
    sess.set_known(edges)
    print("graph_ok", sess.graph_ok(0.001))
    out = []
    for ep in range(2):
        out.append(T.train_epoch(sess, use, usew, 1.0, 0.001, batch_size=24))
        torch.cuda.synchronize()
        tr = sess.trainer
        res[tag + f"_opt{ep}"] = (tr.seg_step.cpu().clone(), tr.exp_avg.cpu().clone(), tr.exp_avg_sq.cpu().clone(), tr.gflat.cpu().clone(), tr.touched.cpu().clone())
    torch.cuda.synchronize()
    if graph:
        recs = sess._graph_state["rec"].cpu().numpy().copy()
        chs = sess._graph_state["chroms"].cpu().numpy().copy()
    else:
        recs = torch.cat(sess._rec_steps[-33:]).cpu().numpy()
        chs = np.asarray(sess._chrom_steps[-33:])
    res[tag + "_rec"] = (recs, chs)
    res[tag] = (out, {n: p.detach().cpu().clone() for n, p in clf.named_parameters()})
T.GRAPH_EPOCHS = True
for x_, y_ in (("e1", "e2"), ("g1", "g2"), ("e1", "g1")):
    rows = []
    for n in res[x_][1]:
        diff = (res[x_][1][n] - res[y_][1][n]).abs().reshape(-1)
        rows.append((float(diff.max()), n))
    rows.sort(reverse=True)
    print(x_, "vs", y_, "recon losses", [round(o[1], 5) for o in res[x_][0]], [round(o[1], 5) for o in res[y_][0]], "| worst:", [(round(m, 6), n.replace("node_embedding.", "")) for m, n in rows[:3]])

a, b = res["e1_rec"][0], res["g1_rec"][0]
bad = [i for i in range(len(a)) if abs(a[i] - b[i]) > 1e-4 * abs(a[i])]
print("epoch-2 steps whose recon loss differs:", bad[:10], "chroms graph", res["g1_rec"][1][:12].tolist(), "eager", res["e1_rec"][1][:12].tolist(), "equal", bool((res["g1_rec"][1] == res["e1_rec"][1]).all()))
for i in bad[:4]:
    print("   step", i, "eager", a[i], "graph", b[i], "chrom", res["g1_rec"][1][i])

for ep in (0, 1):
    a, b = res[f"e1_opt{ep}"], res[f"g1_opt{ep}"]
    print("after epoch", ep, "seg_step equal", torch.equal(a[0], b[0]), "| exp_avg max diff %.2e | exp_avg_sq %.2e | leftover grads eager %.2e graph %.2e | touched equal %s" % (
        float((a[1] - b[1]).abs().max()), float((a[2] - b[2]).abs().max()), float(a[3].abs().max()), float(b[3].abs().max()), torch.equal(a[4], b[4])))
    if not torch.equal(a[0], b[0]):
        d = (a[0] != b[0]).nonzero().reshape(-1).tolist()
        print("   seg_step differs at segments", d[:10], a[0][d[:10]].tolist(), b[0][d[:10]].tolist())
