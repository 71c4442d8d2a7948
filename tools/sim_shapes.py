"""The similarity kernel and the pair pass by shape, default settings: ms per pass from the context's HIP events, EXACT partner steps
per second, Q compared bit for bit with the sequential kernel where that is affordable (m * m * n <= 4e10).
    python tools/sim_shapes.py [m n seed]...        (default: the round-4 shape list)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa
import bench

vhash, dist = SimilarityMatrix.aa()._device_arrays()
shapes = [(1000, 4000, 11), (2000, 10000, 1003), (3583, 7287, 1003), (5000, 5000, 1004), (8000, 3000, 5), (20000, 500, 3), (40000, 300, 4)]
args = [int(x) for x in sys.argv[1:]]
if args:
    shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)]
reps = int(os.environ.get("REPS", "4"))
for m, n, seed in shapes:
    a = synth_msa(m, n, seed)
    rec = {"m": m, "n": n}
    ctx = _lib.Context(0)
    for _ in range(2):
        ctx.upload(a, ord("X"))
        ctx.similarity(vhash, dist)
    ctx.prof_enable(True)
    ctx.lib.msa_prof_reset(ctx.h)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.upload(a, ord("X"))
        mdk, q = ctx.similarity(vhash, dist)
    rec["wall_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    for k in ("sim", "pairs", "encode", "prep", "gaps"):
        ms, cnt = ctx.prof_get(k)
        if cnt:
            rec[k + "_ms"] = round(ms / cnt, 4)
    # partner steps of the pass, EXACT (bench.similarity_w_stream_bytes: per evaluated column and 64-row round the valid rows at or
    # behind the round's first row, in blocks of 16 since round 5 -- the estimate nv^2 / 128 this tool printed through round 4
    # undercounts: a round walks the partner list from its first row on, a column costs ~ nv m / 128 steps)
    wbytes, steps = bench.similarity_w_stream_bytes(a)
    rec["partner_steps"] = steps
    rec["steps_per_s"] = round(steps / (rec["sim_ms"] * 1e-3), 0)
    rec["w_stream_TBs"] = round(wbytes / (rec["sim_ms"] * 1e-3) / 1e12, 2)
    rec["sim_launches"] = int(ctx.last_paths()["sim_launches"])
    rec["sim_waves_per_column"] = int(ctx.last_paths()["sim_waves_per_column"])
    ctx.close()
    if float(m) * m * n <= 4e10 and os.environ.get("CHECK", "1") == "1":
        os.environ["MSA_SIM_KERNEL"] = "seq"
        c2 = _lib.Context(0)
        c2.upload(a, ord("X"))
        _, q2 = c2.similarity(vhash, dist)
        c2.close()
        del os.environ["MSA_SIM_KERNEL"]
        rec["q_equals_sequential_kernel"] = bool(np.array_equal(q.view(np.uint32), q2.view(np.uint32)))
    print(json.dumps(rec), flush=True)
