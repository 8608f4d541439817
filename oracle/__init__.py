"""TEST INFRASTRUCTURE ONLY.

CPU restatement (our own code, plain PyTorch-CPU / numpy) of the reference's
denoising hot path, used as the parity checker by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg.  Nothing under
afford-motion_amd/ (the product) may import this package.

Pinning: the reference has no tests or golden vectors, so the
restatement is pinned against outputs of the reference itself, imported in the
build container by oracle/make_goldens.py (fixtures in tests/golden/).
Exception - "parity unpinned": FPS / kNN (external pointops_cuda, absent) and
the CLIP text encoder (absent); see pointops_ref.py and DESIGN.md.
"""
