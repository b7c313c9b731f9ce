"""INTEGRATION.md section B documents the ctypes stub a MATCHA maintainer would paste into Code/Modules.py.  This test
executes that code block VERBATIM (extracted from the markdown) against the built library: on CPU it checks that the block
compiles and that its struct layouts equal the binding the package itself uses; on the GPU it attaches the stub's
`forward` / `_matcha_descriptors` to a model and compares its logits with the reference's golden vectors (round 1's stub
under-allocated `losses`: an out-of-bounds device write for whoever copied it)."""
import ctypes as C
import os
import re
import types

import numpy as np
import pytest
import torch

from matcha_amd import _lib, synth
from tests.helpers import gold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## B. Keep the reference's `Modules.py` and add the stub"):]
    return re.search(r"```python\n(.*?)```", sec, flags=re.S).group(1)


def _exec_stub():
    os.environ["MATCHA_HIP_LIB"] = _lib.LIB_PATH
    ns = {"torch": torch}
    exec(compile(_stub_source(), "INTEGRATION.md#B", "exec"), ns)
    return ns


def test_stub_compiles_and_its_structs_match_the_package_binding():
    ns = _exec_stub()
    for mine, theirs in ((ns["_Shape"], _lib.Shape), (ns["_Tensors"], _lib.Tensors), (ns["_Frozen"], _lib.Frozen), (ns["_Opts"], _lib.StepOpts)):
        assert C.sizeof(mine) == C.sizeof(theirs)
        assert [f[0] for f in mine._fields_] == [f[0] for f in theirs._fields_]
        assert [getattr(mine, f[0]).offset for f in mine._fields_] == [getattr(theirs, f[0]).offset for f in theirs._fields_]
    assert ns["_TENSOR_FIELDS"] == _lib.TENSOR_FIELDS
    src = _stub_source()
    assert "torch.zeros(3" in src                         # matcha_forward writes losses[3] (include/matcha_hip.h)


@pytest.mark.gpu
def test_stub_forward_reproduces_the_reference_logits():
    from tests.test_hip_model import hip_model
    ns = _exec_stub()
    g = gold("g2_hg38_table_d64.npz")
    clf, _ = hip_model(synth.LAYOUTS["hg38_1mb"], 64, "table", 24)
    clf.eval()
    rt = clf._runtime()                                   # parameters become contiguous fp32 views on the GPU
    clf._matcha_descriptors = types.MethodType(ns["_matcha_descriptors"], clf)
    fwd = types.MethodType(ns["forward"], clf)
    for key in ("k2", "k3", "k5", "mixed"):
        x = torch.from_numpy(g[f"x_{key}"]).cuda()
        lg, rc = fwd(x, return_recon=True)
        ref = g[f"logits_{key}"]
        assert np.abs(lg.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
        assert float(rc[0]) == 0.0
    bad = torch.from_numpy(g["x_k3"]).cuda().clone()
    bad[0, 0] = 10 ** 6
    with pytest.raises(IndexError):
        fwd(bad)
