"""The similarity pass of a synthetic alignment under two settings of the diagnostic switches, contexts alternating (A B A B), Q and MDK
compared bit for bit:   python tools/sim_ab.py "MSA_LG_KSEG=0" "MSA_LG_KSEG=1" m n seed [m n seed ...]      ("" = the defaults)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

vhash, dist = SimilarityMatrix.aa()._device_arrays()
legs = [dict(kv.split("=", 1) for kv in arg.split()) for arg in sys.argv[1:3]]
args = [int(x) for x in sys.argv[3:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)]
GAPPY = os.environ.get("GAPPY")  # rows whose first / last third is gaps (terminal gaps: the list's density varies along the column)
for m, n, seed in shapes:
    a = synth_msa(m, n, seed)
    if GAPPY:
        r = np.random.default_rng(seed)
        rows = r.random(m) < 0.4
        a[rows, : n // 3] = ord("-")
        a[~rows & (r.random(m) < 0.3), 2 * n // 3:] = ord("-")
        blk = slice(m // 5, m // 2)  # a block of rows that is nearly all gaps in half of the columns
        a[blk, ::2] = np.where(r.random((blk.stop - blk.start, (n + 1) // 2)) < 0.9, ord("-"), a[blk, ::2])
    out = [[], []]
    for rnd in range(2):
        for i, env in enumerate(legs):
            saved = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            ctx = _lib.Context(0)
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            ctx.upload(a, ord("X"))
            ctx.similarity(vhash, dist)
            ctx.prof_enable(True)
            ctx.prof_reset()
            for _ in range(3):
                ctx.upload(a, ord("X"))
                mdk, q = ctx.similarity(vhash, dist)
            ms, cnt = ctx.prof_get("sim")
            out[i].append((round(ms / cnt, 3), q.view(np.uint32).copy(), mdk.view(np.uint32).copy(), ctx.last_paths()))
            ctx.close()
    same = all(np.array_equal(r[1], out[0][0][1]) and np.array_equal(r[2], out[0][0][2]) for leg in out for r in leg)
    a_ms, b_ms = min(r[0] for r in out[0]), min(r[0] for r in out[1])
    print(json.dumps({"m": m, "n": n, "gappy": bool(GAPPY), "A": legs[0], "B": legs[1], "sim_ms_A": [r[0] for r in out[0]], "sim_ms_B": [r[0] for r in out[1]],
                      "B_over_A": round(b_ms / a_ms, 3), "launches": [out[0][0][3]["sim_launches"], out[1][0][3]["sim_launches"]],
                      "waves_per_column": out[1][0][3]["sim_waves_per_column"], "bit_identical": bool(same)}), flush=True)
