"""Similarity kernel time (ms) by shape and kernel variant: one column per wave (default: table in LDS, four waves per
workgroup; eight waves per workgroup; table in registers), two columns per wave (q2), the one-grid-per-round
predecessor (bx)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

mat = SimilarityMatrix.aa()
vhash = np.ascontiguousarray(mat._vhash, dtype=np.int32)
dist = np.ascontiguousarray(mat._dist, dtype=np.float32)
shapes = [(2000, 3000), (2000, 5000), (2000, 7000), (2000, 10000), (1000, 4000), (1000, 8000), (4000, 6000), (500, 12000)]
for m, n in shapes:
    a = synth_msa(m, n, 5000 + m + n)
    row = {"m": m, "n": n}
    for label, env in (("lg", {"MSA_SIM_KERNEL": "lg"}), ("lg_8_waves_per_workgroup", {"MSA_SIM_KERNEL": "lg", "MSA_LG_DBG": "2"}),
                       ("lg_table_in_registers", {"MSA_SIM_KERNEL": "lg", "MSA_LG_REGS": "1"}), ("q2", {"MSA_SIM_KERNEL": "q2"}),
                       ("bx", {"MSA_SIM_KERNEL": "bx"})):
        for k in ("MSA_SIM_KERNEL", "MSA_LG_DBG", "MSA_LG_REGS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx = _lib.Context(0)
        ctx.upload(a, ord("X"))
        ctx.similarity(vhash, dist)
        ctx.prof_enable(True)
        for _ in range(3):
            ctx.upload(a, ord("X"))
            ctx.similarity(vhash, dist)
        ms, k = ctx.prof_get("sim")
        row[label] = round(ms / max(k, 1), 3)
        ctx.close()
    print(json.dumps(row), flush=True)
