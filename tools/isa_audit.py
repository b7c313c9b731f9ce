"""Audit gfx950 assembly for VGPR spill code that runs under a STALE EXEC mask (round 5 root cause of the fused_fwd32h fragility).

hipcc (ROCm 7.2) can place a `scratch_store` (a VGPR spill) or `scratch_load` (its reload) at the top of a control-flow JOIN block in
front of the `s_or_b64 exec, exec, sN` that re-enables the lanes which skipped the branch -- when some other instruction (a scalar copy)
was already sitting in front of that `s_or`.  The spill then writes only the lanes that took the branch; if NO lane took it (EXEC = 0:
the branch's condition was wave-uniformly false) nothing is stored and the later reload returns whatever the scratch memory held.

usage:  python tools/isa_audit.py [--zero-scratch NAME ...] file.s [file.s ...]
        (device assembly from `hipcc -S --offload-device-only` or `-save-temps=obj`: the Makefile keeps build/csrc/*gfx950.s and runs this)
--zero-scratch NAME: additionally require `.private_segment_fixed_size: 0` for every kernel whose mangled name contains NAME (the two
fused forward kernels: their register budget is what round 5 fixed).
Flags a scratch access that sits between a block's label and its `s_or_b64 exec, exec, ...` with nothing but scalar instructions or other
spill code in between; exit status 1 if any is found.  (A branch BODY that reloads for its own lanes, computes, and ends in its join is
fine and is not flagged.)"""
import re, sys

RESTORE = re.compile(r"^\s*(s_or_b64|s_mov_b64|s_or_saveexec_b64|s_andn2_saveexec_b64|s_xor_b64|s_andn2_b64)\s+exec\b")


def audit(path):
    bad = 0
    fn, blk, pending, head = None, None, [], True
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", s)
        if m and not s.startswith(".L"):
            fn, blk, pending, head = m.group(1), "entry", [], True
            continue
        if re.match(r"^\.LBB\d+_\d+:", s):
            blk, pending, head = s.split(":")[0], [], True
            continue
        if fn is None or not s or s.startswith(";") or s.startswith("."):
            continue
        if s.startswith("scratch_store") or s.startswith("scratch_load"):
            if head:              # only spill code at the HEAD of a block (nothing but scalar instructions since the label)
                pending.append((ln, s.split(";")[0].strip()))
        elif RESTORE.match(s) and "exec, exec" in s.replace("  ", " ") and s.startswith("s_or_b64"):
            # an EXEC restore: every scratch access seen in this block so far ran under the narrower mask
            for pl, ps in pending:
                print(f"{path}:{pl}: {fn[:70]} {blk}: `{ps}` precedes `{s}` (line {ln})")
                bad += 1
            pending = []
        elif s.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_and_saveexec", "s_barrier")):
            pending, head = [], False          # past the block's head: later restores belong to other regions
        elif s.startswith(("v_", "ds_", "global_", "buffer_", "flat_")):
            pending, head = [], False          # real vector work between the spill and the restore: the block is a branch BODY that ends in its own
                                  # join (reloads for that body's lanes are fine), not spill code stranded in front of a join's restore
    return bad


def scratch_of(path):
    """{kernel name: (private_segment_fixed_size, vgpr_spill_count, sgpr_spill_count)} from the code-object metadata of an assembly file"""
    out = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: +\d+", open(path).read(), re.S):
        b = m.group(0)
        g = lambda k: re.search(r"\.%s: +(\S+)" % k, b).group(1)
        out[g("name")] = (int(g("private_segment_fixed_size")), int(g("vgpr_spill_count")), int(g("sgpr_spill_count")))
    return out


if __name__ == "__main__":
    args, zero = sys.argv[1:], []
    while args and args[0] == "--zero-scratch":
        zero.append(args[1]); args = args[2:]
    n = sum(audit(p) for p in args)
    print(f"{n} scratch access(es) under a stale EXEC mask in {len(args)} file(s)")
    for p in args:
        for name, (scr, vs, ss) in scratch_of(p).items():
            if any(z in name for z in zero) and (scr or vs):
                print(f"{p}: {name}: {scr} bytes of scratch per lane, {vs} spilled VGPRs (must be 0)")
                n += 1
    sys.exit(1 if n else 0)
