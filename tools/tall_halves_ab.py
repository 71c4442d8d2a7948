"""A similarity pass of several launches: its columns as two staggered halves on two streams (MSA_LG_HALVES=2: wherever a pass is several
launches) against one launch sequence (=0), contexts alternating; Q and MDK compared bit for bit; and what the default rule
(msak::lg_halves) picks for the shape.   python tools/tall_halves_ab.py [m n seed]..."""
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
shapes = [(2000, 10000, 1003), (3583, 7287, 1003), (4500, 6000, 4), (3000, 8000, 6), (2000, 5200, 7), (2000, 3000, 2), (5000, 5000, 1004), (6000, 4000, 5),
          (8000, 3000, 5), (4000, 2000, 3), (3000, 1500, 9), (8000, 1500, 6), (9000, 640, 5), (10000, 700, 1), (10000, 500, 1), (12000, 800, 2), (12000, 500, 3),
          (14000, 700, 6), (16000, 600, 2), (20000, 500, 3), (40000, 300, 4)]
args = [int(x) for x in sys.argv[1:]]
if args:
    shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)]
for m, n, seed in shapes:
    a = synth_msa(m, n, seed)
    out = {}
    for rnd in range(2):
        PARTS = os.environ.get("PARTS", "2")  # (MSA_LG_PARTS of the "halves" leg: 2, 3, 4)
        for name, env in (("one sequence", {"MSA_LG_HALVES": "0"}), ("halves", {"MSA_LG_HALVES": "2", "MSA_LG_PARTS": PARTS}), ("default", {})):
            if name == "default" and rnd:
                continue
            os.environ.pop("MSA_LG_HALVES", None)
            os.environ.update(env)
            if os.environ.get("ROUNDS") and name != "default":  # (rounds per launch of both legs: A/B of the cadence under the halves)
                os.environ["MSA_LG_ROUNDS"] = os.environ["ROUNDS"]
            ctx = _lib.Context(0)
            os.environ.pop("MSA_LG_HALVES", None)
            os.environ.pop("MSA_LG_ROUNDS", None)
            os.environ.pop("MSA_LG_PARTS", None)
            ctx.upload(a, ord("X"))
            mdk, q = ctx.similarity(vhash, dist)
            ctx.prof_enable(True)
            ctx.prof_reset()
            t0 = time.perf_counter()
            for _ in range(3):
                ctx.upload(a, ord("X"))
                mdk, q = ctx.similarity(vhash, dist)
            wall = (time.perf_counter() - t0) / 3 * 1e3
            ms, cnt = ctx.prof_get("sim")
            out.setdefault(name, []).append((round(ms / cnt, 3), round(wall, 3), q.view(np.uint32).copy(), mdk.view(np.uint32).copy(), ctx.last_paths()))
            ctx.close()
    base = out["one sequence"][0]
    same = all(np.array_equal(r[2], base[2]) and np.array_equal(r[3], base[3]) for rs in out.values() for r in rs)
    print(json.dumps({"m": m, "n": n, "sim_ms_one_sequence": [r[0] for r in out["one sequence"]], "sim_ms_halves": [r[0] for r in out["halves"]],
                      "wall_ms_one_sequence": [r[1] for r in out["one sequence"]], "wall_ms_halves": [r[1] for r in out["halves"]],
                      "launches": [out["one sequence"][0][4]["sim_launches"], out["halves"][0][4]["sim_launches"]],
                      "default_rule_launches": out["default"][0][4]["sim_launches"], "sim_ms_default": out["default"][0][0],
                      "waves_per_column": out["halves"][0][4]["sim_waves_per_column"], "parts": int(os.environ.get("PARTS", "2")), "q_and_mdk_bit_identical": bool(same)}), flush=True)
