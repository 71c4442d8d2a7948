"""Where the first trim of a fresh Alignment spends its time (bench.py's `value_cold`): the steps of `trimmer.trim` timed one by one on
fresh objects, against the same steps on an alignment that has been trimmed before.   python tools/cold_trim.py [m n]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import Alignment, AutomaticTrimmer, _lib
from pytrimal_amd.synth import synth_msa

m, n = (int(x) for x in (sys.argv[1:3] if len(sys.argv) > 2 else (2000, 10000)))
a = synth_msa(m, n, 1003)
names = [b"s%d" % i for i in range(m)]
rows = [bytes(r) for r in a]
tr = AutomaticTrimmer("automated1", platform="hip")
warm = Alignment(names, rows)
for _ in range(3):
    tr.trim(warm)


def steps(ali):
    t = [time.perf_counter()]
    nm, dense, indet, params, keep = tr._prepare(ali)
    t.append(time.perf_counter())
    ctx = _lib.thread_context()
    ctx.upload(dense, indet, wait=False)
    t.append(time.perf_counter())
    keep_res, keep_seq, info = ctx.trim(params)
    t.append(time.perf_counter())
    rows_ = ctx.only_gaps_rows() if info.warnings else []
    out = tr._finish(nm, dense, ali._datatype, keep_res, keep_seq, info, rows_, ctx.gaps_cached(0), params)
    t.append(time.perf_counter())
    return [round((b - a_) * 1e3, 3) for a_, b in zip(t, t[1:])]


cold = [steps(Alignment(names, rows)) for _ in range(5)]
hot = [steps(warm) for _ in range(5)]
print(json.dumps({"m": m, "n": n, "steps": ["prepare", "upload (enqueue)", "trim (waits for everything)", "finish"],
                  "fresh_alignment_ms": np.median(np.array(cold), axis=0).round(3).tolist(),
                  "trimmed_before_ms": np.median(np.array(hot), axis=0).round(3).tolist()}))
