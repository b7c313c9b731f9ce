"""The compiler-bug guard of round 5 as a test (tools/isa_audit.py).

hipcc of ROCm 7.2 can strand a VGPR spill at the head of a control-flow join block IN FRONT of the `s_or_b64 exec` that re-enables the
lanes which skipped the branch; when the branch was skipped by the whole wavefront (EXEC = 0) the spill stores nothing and the reload
returns uninitialised scratch -- that is what made round 4's fused_fwd32h_kernel lose its ddyn0 / dXs rows when unreachable code was
appended to it.  The Makefile keeps every translation unit's device assembly and refuses to link a library that contains the pattern;
this test (a) pins the detector on a hand-written positive and a negative, (b) runs it over the assembly of the shipped build and
(c) requires the two fused forward kernels to compile without scratch."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_audit  # noqa: E402

BAD = """
_Z6kernelv:
\ts_and_saveexec_b64 s[20:21], s[24:25]
\ts_cbranch_execz .LBB0_2
.LBB0_1:
\tglobal_store_dwordx4 v[130:131], v[34:37], off
.LBB0_2:
\ts_mov_b64 s[30:31], s[28:29]
\tscratch_store_dwordx2 off, v[204:205], off offset:4 ; 8-byte Folded Spill
\ts_or_b64 exec, exec, s[20:21]
\tv_mov_b32_e32 v130, v34
\ts_endpgm
"""
# a branch BODY that reloads for its own lanes, computes, spills the update and ends in its join: legitimate
GOOD = """
_Z6kernelv:
\ts_and_saveexec_b64 s[70:71], s[4:5]
\ts_cbranch_execz .LBB0_2
.LBB0_1:
\tscratch_load_dwordx2 v[48:49], off, off offset:8 ; 8-byte Folded Reload
\ts_waitcnt vmcnt(0)
\tv_fmac_f32_e32 v34, v32, v49
\tscratch_store_dwordx2 off, v[48:49], off offset:8 ; 8-byte Folded Spill
\ts_or_b64 exec, exec, s[70:71]
.LBB0_2:
\ts_or_b64 exec, exec, s[70:71]
\ts_endpgm
"""


def test_detector_flags_the_stranded_spill_and_not_a_branch_body(tmp_path, capsys):
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(BAD)
    good.write_text(GOOD)
    assert isa_audit.audit(str(bad)) == 1
    assert isa_audit.audit(str(good)) == 0
    assert "precedes `s_or_b64 exec, exec, s[20:21]`" in capsys.readouterr().out


def test_shipped_build_has_no_stranded_spill_and_the_forward_kernels_no_scratch():
    csrc = os.path.join(ROOT, "matcha_amd", "csrc")
    asm = sorted(glob.glob(os.path.join(ROOT, "build", "csrc", "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if len(asm) < len(glob.glob(os.path.join(csrc, "*.hip"))):
        # no build directory here (a fresh checkout): build the library, which leaves the assembly behind
        r = subprocess.run(["make", "-C", csrc, "-j", str(min(8, os.cpu_count() or 1))], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        asm = sorted(glob.glob(os.path.join(ROOT, "build", "csrc", "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    assert len(asm) == len(glob.glob(os.path.join(csrc, "*.hip")))
    assert sum(isa_audit.audit(p) for p in asm) == 0
    fwd = isa_audit.scratch_of([p for p in asm if os.path.basename(p).startswith("fused_fwd32-")][0])
    kernels = {k: v for k, v in fwd.items() if "fused_fwd32_kernel" in k or "fused_fwd32h_kernel" in k}
    assert len(kernels) == 12                                   # six ML instances of each
    for name, (scratch, vgpr_spills, _) in kernels.items():
        assert scratch == 0 and vgpr_spills == 0, (name, scratch, vgpr_spills)
