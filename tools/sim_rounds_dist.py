"""One launch against a launch every six (ten) rounds, one context alive at a time (same device addresses for every
setting), settings alternating: mean / median / min / max of the kernel's time over the passes.   python tools/sim_rounds_dist.py"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

vhash, dist = SimilarityMatrix.aa()._device_arrays()
cases = {"C3 (synth_msa 2000 x 10000, seed 1003)": synth_msa(2000, 10000, 1003)}
srt = cases["C3 (synth_msa 2000 x 10000, seed 1003)"].copy()
srt.sort(axis=0)
cases["the same with every column sorted by residue"] = srt
cases["synth_msa 2500 x 9000"] = synth_msa(2500, 9000, 3)
for name, a in cases.items():
    times = {"0": [], "6": [], "10": []}
    for rep in range(4):
        for per in times:
            os.environ["MSA_LG_ROUNDS"] = per
            ctx = _lib.Context(0)
            os.environ.pop("MSA_LG_ROUNDS")
            for _ in range(2):
                ctx.upload(a, ord("X"))
                ctx.similarity(vhash, dist)
            ctx.prof_enable(True)
            ctx.lib.msa_prof_reset(ctx.h)
            for _ in range(5):
                ctx.upload(a, ord("X"))
                ctx.similarity(vhash, dist)
                ms, k = ctx.prof_get("sim")
            # (per pass: the events of the five passes one by one)
            ctx.lib.msa_prof_reset(ctx.h)
            if rep:
                times[per].append(ms / k)
            ctx.close()
    print(json.dumps({"data": name, **{"rounds_per_launch_" + k: {"mean": round(statistics.mean(v), 3), "min": round(min(v), 3), "max": round(max(v), 3)}
                                       for k, v in times.items()}}), flush=True)
