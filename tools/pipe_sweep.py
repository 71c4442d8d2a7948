"""The similarity pass of synthetic alignments with the waves-per-column forced (MSA_LG_SPLIT) under the pipelined split kernel
(MSA_LG_PIPE=2) against the barrier kernel at its default split (MSA_LG_PIPE=0) and the shipped default:
   python tools/pipe_sweep.py "2 3 4 6 8 12" m n seed [m n seed ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

vhash, dist = SimilarityMatrix.aa()._device_arrays()
splits = [int(x) for x in sys.argv[1].split()]
args = [int(x) for x in sys.argv[2:]]


def run(a, env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    ctx = _lib.Context(0)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    ctx.upload(a, ord("X")); ctx.similarity(vhash, dist)
    ctx.prof_enable(True); ctx.prof_reset()
    best = None
    for _ in range(2):
        ctx.prof_reset()
        for _ in range(3):
            ctx.upload(a, ord("X")); mdk, q = ctx.similarity(vhash, dist)
        ms, cnt = ctx.prof_get("sim")
        best = ms / cnt if best is None else min(best, ms / cnt)
    P = ctx.last_paths()
    ctx.close()
    return round(best, 3), q.view(np.uint32).copy(), int(P["sim_waves_per_column"]), int(P["sim_launches"])


for m, n, seed in [tuple(args[i:i + 3]) for i in range(0, len(args), 3)]:
    a = synth_msa(m, n, seed)
    base_ms, q0, s0, l0 = run(a, {"MSA_LG_PIPE": "0"})
    def_ms, q1, s1, l1 = run(a, {})
    rec = {"m": m, "n": n, "barrier": {"ms": base_ms, "S": s0, "launches": l0}, "default": {"ms": def_ms, "S": s1, "launches": l1}, "pipe": {}, "bit_identical": bool(np.array_equal(q0, q1))}
    for S in splits:
        ms, q, s, l = run(a, {"MSA_LG_PIPE": "2", "MSA_LG_SPLIT": str(S)})
        rec["pipe"][str(S)] = ms
        rec["bit_identical"] = rec["bit_identical"] and bool(np.array_equal(q0, q))
    print(json.dumps(rec), flush=True)
