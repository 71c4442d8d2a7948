"""Kernel times of one small trim (C1 = ENOG 209 x 1227 strictplus): the context's own HIP-event profile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from pytrimal_amd import Alignment, AutomaticTrimmer, _lib

ali = Alignment.load(os.path.join(ROOT, "tests/golden/data/ENOG411BWBU.seq40.res60.fasta"))
tr = AutomaticTrimmer("strictplus", platform="hip")
for _ in range(3): tr.trim(ali)
ts = []
for _ in range(50):
    t = time.perf_counter(); tr.trim(ali); ts.append(time.perf_counter() - t)
ts.sort()
print("trim ms: median %.3f min %.3f" % (ts[25] * 1e3, ts[0] * 1e3))
ctx = _lib.thread_context()
ctx.prof_enable(True); ctx.prof_reset()
for _ in range(10): tr.trim(ali)
for nm in ("prep", "pairs", "idstats", "gaps", "encode", "sim", "simnum", "simden", "overlap"):
    ms, k = ctx.prof_get(nm)
    if k: print("  %-8s %.1f us x %d" % (nm, ms / k * 1e3, k // 10))
