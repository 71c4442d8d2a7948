"""Time of the pair-count pass (identity + weight matrices): the software-pipelined loop and the triangle-only grid against the plain ones."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.synth import synth_msa
for (m, n, seed) in ((5000, 5000, 1004), (2000, 10000, 1003), (1000, 4000, 2000), (500, 2000, 1002), (3000, 3000, 5), (8000, 2000, 6)):
    a = synth_msa(m, n, seed)
    for ti, xcd, pipe, dense in (("8", "1", "1", "1"), ("8", "1", "1", "0"), ("8", "0", "0", "0"), ("8", "1", "1", "1"), ("8", "1", "1", "0")):
        os.environ["MSA_PAIR_TI"] = ti
        os.environ["MSA_PAIR_XCD"] = xcd
        os.environ["MSA_PAIR_PIPE"] = pipe
        os.environ["MSA_PAIR_DENSE"] = dense
        ctx = _lib.Context(0)
        ctx.upload(a, ord("X")); ctx.identity_stats()
        ctx.prof_enable(True)
        for _ in range(5):
            ctx.upload(a, ord("X")); ctx.identity_stats()
        ms, k = ctx.prof_get("pairs")
        ms_prep, k_prep = ctx.prof_get("prep")
        ms_gaps, k_gaps = ctx.prof_get("gaps")
        print(json.dumps({"m": m, "n": n, "TI": int(ti), "triangle_grid": int(xcd), "pipelined": int(pipe), "dense_codes": int(dense), "pairs_ms": round(ms / k, 4), "prep_ms": round(ms_prep / max(k_prep, 1), 4), "gaps_ms": round(ms_gaps / max(k_gaps, 1), 4), "pair_cols_per_s": round(m * (m - 1) / 2 * n / (ms / k * 1e-3), 1)}), flush=True)
        ctx.close()
