"""Cross-check of the similarity kernels on random shapes (no oracle: the kernels are independent implementations of the
same bit-exact statistic -- per-lane grids / one grid per round / dependent-add chains -- so any disagreement is a bug).
Usage: python tools/cross_check.py [cases] [seed]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mat = SimilarityMatrix.aa()
vhash = np.ascontiguousarray(mat._vhash, dtype=np.int32)
dist = np.ascontiguousarray(mat._dist, dtype=np.float32)
ALPHA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
r = np.random.default_rng(seed)


def make(m, n):
    kind = r.integers(0, 5)
    a = ALPHA[r.integers(0, 20, (m, n))].copy()
    if kind == 0:      # conserved columns with a few substitutions
        a[:] = ALPHA[r.integers(0, 20, n)][None, :]
        sub = r.random((m, n)) < r.choice([0.001, 0.02, 0.2])
        a[sub] = ALPHA[r.integers(0, 20, int(sub.sum()))]
    g = r.choice([0.0, 0.05, 0.3, 0.6, 0.79]) if kind != 1 else r.random(n)[None, :] * 0.95
    a[r.random((m, n)) < g] = ord("-")
    if kind == 2:      # blocks of gap rows, identical sequences
        a[m // 3:m // 3 + r.integers(1, max(2, m // 2))] = ord("-")
        a[1] = a[0]
    if kind == 3:      # sorted columns
        a[:, : n // 2] = np.sort(a[:, : n // 2], axis=0)
    a[r.random((m, n)) < 0.01] = ord("X")
    return np.ascontiguousarray(a)


def run(kernel, a):
    for k in ("MSA_SIM_KERNEL", "MSA_LG_BIG", "MSA_LG_ROUNDS", "MSA_LG_SPLIT"):
        os.environ.pop(k, None)
    if kernel == "seq":
        os.environ["MSA_SIM_KERNEL"] = "seq"
    elif kernel == "lg-big":
        os.environ["MSA_LG_BIG"] = "1"
    elif kernel == "lg-rounds":  # one round per launch, the columns' state through memory (by itself: six from 1800 rows on)
        os.environ["MSA_LG_ROUNDS"] = "1"
    elif kernel.startswith("lg-split-"):  # a workgroup of S waves per column (by itself: tall alignments)
        os.environ["MSA_LG_SPLIT"] = kernel.rsplit("-", 1)[1]
    ctx = _lib.Context(0)
    try:
        ctx.upload(a, ord("X"))
        mdk, q = ctx.similarity(vhash, dist)
        return mdk.view(np.uint32).copy(), q.view(np.uint32).copy()
    finally:
        ctx.close()


bad = 0
for i in range(cases):
    m = int(r.choice([2, 3, 17, 63, 64, 65, 127, 129, 200, 500, 1000, 1500, 2100, 3000]))
    n = int(r.choice([1, 5, 64, 100, 257, 600]))
    if m >= 1500:
        n = min(n, 100)
    a = make(m, n)
    ref = run("seq", a)  # the reference's two loops, one lane per column
    for k in ("lg", "lg-big", "lg-rounds", "lg-split-2", "lg-split-5", "lg-split-16"):
        got = run(k, a)
        ok = np.array_equal(got[1], ref[1]) and np.array_equal(got[0], ref[0])
        if not ok:
            bad += 1
            d = np.nonzero(got[1] != ref[1])[0]
            print(json.dumps({"case": i, "m": m, "n": n, "kernel": k, "columns_differing": int(d.size), "first": d[:5].tolist()}), flush=True)
print(json.dumps({"cases": cases, "seed": seed, "mismatches": bad}))
sys.exit(1 if bad else 0)
