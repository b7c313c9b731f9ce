"""Sanitizers on the library's host logic (SURVEY.md §5 "race detection / sanitizers"): GPU
AddressSanitizer is not available on the MI355X pool, so the NON-KERNEL code of libmatcha_hip -- argument validation, workspace
sizing and carving, the option table -- is compiled for the host only with -fsanitize=address,undefined and driven by
tests/host/host_checks.hip, together with the plain-C restatement of the ragged plan (oracle/c/ragged_plan.c) under a fuzzer with
its invariants.  Runs in the CPU container; no kernel is launched (launch attempts must come back as MATCHA_EHIP)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_under_asan_ubsan():
    csrc = os.path.join(ROOT, "matcha_amd", "csrc")
    build = subprocess.run(["make", "-C", csrc, "-j", str(min(8, os.cpu_count() or 1)), "host-asan"], capture_output=True, text=True)
    assert build.returncode == 0, build.stdout[-3000:] + build.stderr[-3000:]
    env = dict(os.environ, MATCHA_FUSED_DBG="3", ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for k in list(env):
        if k.startswith("MATCHA_DISABLE"):
            env.pop(k)
    run = subprocess.run([os.path.join(ROOT, "build", "host_asan", "host_checks")], capture_output=True, text=True, env=env, timeout=600)
    out = run.stdout + run.stderr
    assert run.returncode == 0, out[-4000:]
    assert "ALL HOST CHECKS PASSED" in out
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
