"""CPU oracle for the MATCHA hyperedge-classifier training path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``matcha_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker /
reported baseline.  The product path is the HIP library
(``matcha_amd/lib/libmatcha_hip.so``) and fails loudly when it is missing.

Parity status (see DESIGN.md "Oracle"):
  * model forward / backward / AdamW / embeddings: PINNED against outputs of
    the real reference (``/root/reference/Code/Modules.py`` imported in the
    build container by ``tests/golden/make_golden.py``; fixtures committed
    under ``tests/golden/``).
  * negative sampler: parity UNPINNED at the ``pybloom_live`` boundary (the
    package is a third-party, un-vendored, un-versioned dependency of the
    reference, absent from this image); pinned by the invariants of
    SURVEY.md §8(c3) plus statistics captured from the reference's own
    ``generate_negative`` run with an exact-set stand-in.
"""
