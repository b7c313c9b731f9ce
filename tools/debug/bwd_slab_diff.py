"""Which (head, chunk) workgroups of fused_bwdh_kernel produce different slabs in two identical runs, and which part of the slab?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from matcha_amd import synth
from matcha_amd.engine import Trainer
from tests.test_hip_model import hip_model
from tests.test_hip_properties import _big_batch

num = synth.LAYOUTS["hg38_1mb"]; N = int(np.sum(num))
rng = np.random.default_rng(1)
ks = [int(k) for k in os.environ.get("KS", "2,3,4,5").split(",")]
xs = [np.pad(synth.make_edges_fast(rng, N, k, 16384 // len(ks)), ((0, 0), (0, 5 - k))) for k in ks]
xx = np.concatenate(xs)
x = torch.from_numpy(xx[rng.permutation(len(xx))]).cuda()
y = (torch.rand(len(x), device="cuda") < 0.25).float()
w = torch.ones(len(x), device="cuda")
S = 2 * 4096 + 4 * 64
slabs = []
for run in range(3):
    clf, _ = hip_model(num, 64, "table", 3); clf.train()
    tr = Trainer(clf, base_seed=5)
    tr.forward_backward(x, y.reshape(-1), w.reshape(-1), 1.0, 0.001, 0)
    torch.cuda.synchronize()
    B, L = x.shape
    ws, _ = tr._buffers(B, L)
    buf = C.create_string_buffer(8192)
    fn = tr.lib.matcha_debug_layout; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_char_p, C.c_size_t]
    assert fn(C.byref(tr.rt.shape), B, L, buf, 8192) == 0
    lay = dict((l.split()[0], int(l.split()[1])) for l in buf.value.decode().strip().split("\n"))
    off = lay["fb_ws"]
    slabs.append(ws[off:off + 8 * 64 * S * 4].view(torch.float32).cpu().numpy().reshape(8, 64, S).copy())
    qk = lay["qkv_records"]
for r in (1, 2):
    d = slabs[r] != slabs[0]
    print("run", r, "differing workgroups (head, chunk): dB dM db dbdyn dxpad")
    for h in range(8):
        for c in range(64):
            if d[h, c].any():
                print("  ", h, c, int(d[h, c, :4096].sum()), int(d[h, c, 4096:8192].sum()), int(d[h, c, 8192:8256].sum()), int(d[h, c, 8256:8320].sum()), int(d[h, c, 8320:8384].sum()), "probe", int(d[h, c, 8384:8448].sum()),
                      "max rel", float(np.abs(slabs[r][h, c] - slabs[0][h, c]).max() / (np.abs(slabs[0][h, c]).max() + 1e-30)))
