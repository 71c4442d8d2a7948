"""Seeded synthetic protein MSAs for the BASELINE.json configurations.

The draw order is part of the contract: the same (m, n, seed) must give the same bytes on the
build container and on the GPU box, because inputs ship as seeds, not files.
"""
import numpy as np

ALPHA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)

# BASELINE.json `configs` -> (m, n, seed); C5 is 64 alignments with seeds 2000..2063
CONFIGS = {
    "C2": (500, 2000, 1002),
    "C3": (2000, 10000, 1003),
    "C4": (5000, 5000, 1004),
    "C5": (1000, 4000, 2000),
}


def synth_msa(m, n, seed):
    """Return a C-contiguous uint8 [m, n] residue matrix (raw ASCII bytes)."""
    r = np.random.default_rng(seed)
    root = ALPHA[r.integers(0, 20, n)]
    p = r.beta(2, 2, n)
    copy = r.random((m, n)) < p
    a = np.where(copy, root[None, :], ALPHA[r.integers(0, 20, (m, n))])
    g = r.beta(0.6, 1.8, n)
    forced = r.random(n) < 0.05
    g[forced] = r.uniform(0.85, 0.99, n)[forced]
    a[r.random((m, n)) < g[None, :]] = ord("-")
    a[(r.random((m, n)) < 0.005) & (a != ord("-"))] = ord("X")
    return np.ascontiguousarray(a, dtype=np.uint8)
