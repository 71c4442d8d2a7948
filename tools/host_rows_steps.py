"""A few trims from page-locked host rows (upload without waiting + msa_trim), for a rocprofv3 timeline of the upload
slabs against the pair pass:   rocprofv3 --kernel-trace --memory-copy-trace ... -- python3 tools/host_rows_steps.py [C3|C4] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import AutomaticTrimmer, RepresentativeTrimmer, Alignment, _lib
from pytrimal_amd.synth import synth_msa

which = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if which == "C3":
    a = synth_msa(2000, 10000, 1003)
    tr = AutomaticTrimmer("automated1", platform="hip")
else:
    a = synth_msa(5000, 5000, 1004)
    tr = RepresentativeTrimmer(identity_threshold=0.5, platform="hip")
ali = Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a])
names, dense, indet, params, keep = tr._prepare(ali)
ctx = _lib.thread_context()
for k in range(steps):
    t = time.perf_counter()
    ctx.upload(a, indet, pin=True, wait=False)
    ctx.trim(params)
    print("step %d: %.3f ms" % (k, (time.perf_counter() - t) * 1e3), flush=True)
