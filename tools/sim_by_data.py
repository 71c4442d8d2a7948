"""The similarity kernel on data that is not `synth_msa`: C3-sized inputs (2000 x 10000) built from
the reference's ENOG411BWBU fixture and from adversarial constructions.  Per case: kernel ms (HIP events), ordered rows
per column (= mispredicted + binade-crossing rows: what the per-lane grid predictor costs), rounds, the cycle split of a
wave (from a stamped run as one launch) -- and the result of the kernel as shipped (a launch every six rounds) compared bit for
bit with the plain sequential kernel (MSA_SIM_KERNEL=seq).
   python tools/sim_by_data.py > profiles/r03_sim_by_data.jsonl"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa: F401
from bx_stamps import stamped_similarity
from pytrimal_amd import Alignment, _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

M, N = 2000, 10000
ALPHA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
r = np.random.default_rng(2026)


def mutate(a, rate):
    sub = (r.random(a.shape) < rate) & (a != ord("-"))
    a = a.copy()
    a[sub] = ALPHA[r.integers(0, 20, int(sub.sum()))]
    return a


def enog_like(rate):
    """rows resampled from ENOG411BWBU (209 x 1227: 40 % gaps in row-correlated runs, no X), its columns tiled to N,
    every copy mutated at `rate` per residue"""
    ali = Alignment.load(os.path.join(ROOT, "tests", "golden", "data", "ENOG411BWBU.seq40.res60.fasta"))
    base = ali._matrix
    rows = base[r.integers(0, base.shape[0], M)]
    cols = np.concatenate([np.arange(base.shape[1])] * (N // base.shape[1] + 1))[:N]
    return np.ascontiguousarray(mutate(rows[:, cols], rate))


def blocky():
    """two families of 1000 sequences, 5 % divergence inside a family, unrelated between them; 5 % gaps"""
    roots = ALPHA[r.integers(0, 20, (2, N))]
    a = np.repeat(roots, M // 2, axis=0)
    a = mutate(a, 0.05)
    a[r.random(a.shape) < 0.05] = ord("-")
    return np.ascontiguousarray(a)


def gappy(rate):
    a = synth_msa(M, N, 77)
    a = np.where(a == ord("-"), ALPHA[r.integers(0, 20, a.shape)], a)
    a[r.random(a.shape) < rate] = ord("-")
    return np.ascontiguousarray(a.astype(np.uint8))


def sorted_columns():
    a = synth_msa(M, N, 1003)
    return np.ascontiguousarray(np.sort(a, axis=0))


CASES = [("synth_msa seed 1003 (the bench workload)", lambda: synth_msa(M, N, 1003)),
         ("ENOG411BWBU resampled to 2000 x 10000, 10 % mutations", lambda: enog_like(0.10)),
         ("ENOG411BWBU resampled, 30 % mutations", lambda: enog_like(0.30)),
         ("two families of 1000 sequences (blocky)", blocky),
         ("75 % gaps in every column, independent", lambda: gappy(0.75)),
         ("2 % gaps", lambda: gappy(0.02)),
         ("synth_msa with every column sorted by residue", sorted_columns)]
mat = SimilarityMatrix.aa()
vhash, dist = mat._device_arrays()
for name, make in CASES:
    a = make()
    mdk, q, rec = stamped_similarity(a)  # (stamped, as ONE launch: ordered rows, rounds, cycle split per column)
    rec["sim_ms_stamped_one_launch"] = rec.pop("sim_ms")
    # the kernel as shipped (a launch every six rounds), unstamped: mean of six passes behind two untimed ones
    ctx = _lib.Context(0)
    for _ in range(2):
        ctx.upload(a, ord("X"))
        ctx.similarity(vhash, dist)
    ctx.prof_enable(True)
    ctx.lib.msa_prof_reset(ctx.h)
    for _ in range(6):
        ctx.upload(a, ord("X"))
        mdk, q = ctx.similarity(vhash, dist)
    ms, k = ctx.prof_get("sim")
    rec["sim_ms"] = round(ms / k, 3)
    ctx.close()
    os.environ["MSA_SIM_KERNEL"] = "seq"
    seq = _lib.Context(0)
    os.environ.pop("MSA_SIM_KERNEL")
    seq.upload(a, ord("X"))
    avg, mx = seq.identity_stats()
    mdk2, q2 = seq.similarity(vhash, dist)
    seq.close()
    gaps = (a == ord("-")).mean()
    evaluated = int((((a == ord("-")).sum(axis=0) / np.float32(M)) < np.float32(0.8)).sum())
    print(json.dumps({"data": name, "m": M, "n": N, "gap_fraction": round(float(gaps), 3), "avg_identity": round(float(avg), 4),
                      "columns_evaluated": evaluated, "bit_identical_to_sequential_kernel": bool(np.array_equal(q.view(np.uint32), q2.view(np.uint32))
                                                                                               and np.array_equal(mdk.view(np.uint32), mdk2.view(np.uint32))),
                      **rec}), flush=True)
