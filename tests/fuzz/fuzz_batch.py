"""Randomised check of `msa_trim_batch`: random batches (2 .. 300 alignments, shapes 2 x 1 .. 700 x 1500, random compositions,
trimmers of every kind mixed per batch through one parameter block each) through `_lib.Batch.trim` -- the engine for the small
alignments whose trim it takes, the worker contexts for the rest -- against the oracle's trim, alignment by alignment: masks,
return codes, the rows behind the gaps-only warning.   python tests/fuzz/fuzz_batch.py [seconds] [seed]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import oracle  # noqa: E402
from pytrimal_amd import _lib  # noqa: E402
from pytrimal_amd.matrix import SimilarityMatrix  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
mx = SimilarityMatrix.aa()
vhash, dist = mx._device_arrays()


def alignment():
    m = int(rng.choice([2, 3, 7, 21, 40, 64, 65, 100, 128, 129, 130, 300, 700])) + int(rng.integers(0, 5))
    n = int(rng.choice([1, 5, 31, 33, 64, 100, 257, 600, 1500])) + int(rng.integers(0, 7))
    keep = float(rng.choice([0.2, 0.45, 0.6, 0.7, 0.85, 0.97]))
    a = AA[rng.integers(0, 20, (m, n))].copy()
    if rng.random() < 0.5:  # a family: most rows close to the first
        a[:] = a[0]
        sub = rng.random((m, n)) < rng.choice([0.02, 0.2, 0.5])
        a[sub] = AA[rng.integers(0, 20, int(sub.sum()))]
    a[rng.random((m, n)) > keep] = ord("-")
    if rng.random() < 0.3:  # rows that may end up with gaps only
        a[rng.integers(0, m, max(1, m // 8)), :] = ord("-")
        a[0, : max(1, n // 40)] = ord("A")
    if rng.random() < 0.1:
        a[rng.integers(0, m), rng.integers(0, n)] = ord("O")  # not in the matrix: a similarity trim must raise
    if rng.random() < 0.2 and m > 3:  # duplicated rows
        a[rng.integers(0, m)] = a[rng.integers(0, m)]
    return np.ascontiguousarray(a)


METHODS = ["strict", "strictplus", "automated1", "gappyout", "nogaps", "noallgaps"]


def settings(m=6):
    r = rng.random()
    if r < 0.6:
        return dict(method=str(rng.choice(METHODS)))
    if r < 0.7:
        return dict(gap_threshold=float(rng.choice([0.3, 0.6, 0.9])))
    if r < 0.8:
        return dict(similarity_threshold=float(rng.choice([0.1, 0.4])), gap_threshold=float(rng.choice([0.5, 0.8])),
                    conservation_percentage=float(rng.choice([20, 60])))
    if r < 0.84:
        return dict(method="strict", window=int(rng.integers(1, 4)))  # (a gap window in front of the similarity pipeline: not the engine's)
    if r < 0.87:
        return dict(gap_threshold=float(rng.choice([0.3, 0.8])), gap_window=int(rng.integers(1, 3)))  # (... on a gap-only trim: the engine's)
    if r < 0.92:
        return dict(residue_overlap=float(rng.choice([0.3, 0.6, 0.9])), sequence_overlap=float(rng.choice([20.0, 50.0, 80.0])))
    if r < 0.95:
        return dict(method="noduplicateseqs")
    if r < 0.98:
        return dict(identity_threshold=float(rng.choice([0.3, 0.6])))
    return dict(clusters=int(rng.integers(1, min(m, 6) + 1)))


def params_for(kw):
    P = _lib.TrimParams(0, -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, vhash.ctypes.data, dist.ctypes.data, len(mx))
    if "method" in kw:
        P.method = _lib.METHOD_CODES[kw["method"]]
    if "gap_threshold" in kw:
        P.gap_threshold = float(np.float32(1) - np.float32(kw["gap_threshold"]))
    if "similarity_threshold" in kw:
        P.similarity_threshold = kw["similarity_threshold"]
    if "conservation_percentage" in kw:
        P.conservation_percentage = kw["conservation_percentage"]
    if "window" in kw:
        P.window = kw["window"]
    if "gap_window" in kw:
        P.gap_window = kw["gap_window"]
    if "clusters" in kw:
        P.clusters = kw["clusters"]
    if "residue_overlap" in kw:
        P.residue_overlap, P.sequence_overlap = kw["residue_overlap"], kw["sequence_overlap"]
    if "identity_threshold" in kw:
        P.max_identity = kw["identity_threshold"]
    return P


# two batch objects: the shipped policy (fewer than 40 eligible alignments go to the worker contexts, each a compact pipeline), and
# the batched-kernel engine for any number of them (the library reads the switch when a batch object is created)
batch_default = _lib.Batch(0, 3)
os.environ["MSA_BATCH_ENGINE_MIN"] = "1"
batch_engine = _lib.Batch(0, 3)
os.environ.pop("MSA_BATCH_ENGINE_MIN")
t_end = time.time() + budget
batches = cases = raised = engine_like = 0
failures = []
while time.time() < t_end and not failures:
    count = int(rng.choice([2, 5, 17, 60, 300]))
    if count > 60:
        items = [(a, settings(a.shape[0])) for a in (alignment() for _ in range(12))]
        items = [items[int(i)] for i in rng.integers(0, 12, count)]  # (many alignments, few distinct ones: the oracle is the slow side)
    else:
        items = [(a, settings(a.shape[0])) for a in (alignment() for _ in range(count))]
    batch = batch_engine if rng.random() < 0.5 else batch_default
    out = batch.trim([(a, ord("X"), params_for(kw)) for a, kw in items])
    memo = {}
    for k, ((a, kw), (res, seq, info, rc, rows)) in enumerate(zip(items, out)):
        key = (id(a), json.dumps(kw, sort_keys=True))
        if key not in memo:
            try:
                memo[key] = oracle.trim(a, matrix=oracle.aa_matrix(), indet=ord("X"), **kw)
            except oracle.OracleError as e:
                memo[key] = e
        want = memo[key]
        cases += 1
        if isinstance(want, oracle.OracleError):
            raised += 1
            if rc == _lib.OK:
                failures.append({"batch": batches, "k": k, "shape": list(a.shape), "settings": kw, "oracle": repr(want), "rc": rc})
            continue
        ores, oseq, _ = want
        if rc != _lib.OK or not np.array_equal(res, ores.astype(bool)) or not np.array_equal(seq, oseq.astype(bool)):
            failures.append({"batch": batches, "k": k, "shape": list(a.shape), "settings": kw, "rc": rc})
    batches += 1
batch_default.close()
batch_engine.close()
print(json.dumps({"mismatch": bool(failures), "failures": failures[:5], "batches": batches, "alignments": cases,
                  "alignments_where_both_raise": raised, "seconds": round(budget, 1), "seed": seed}))
sys.exit(1 if failures else 0)
