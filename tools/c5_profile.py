"""Where does a 1000 x 4000 trim() spend its time (host rows -> masks)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import numpy as np
from pytrimal_amd import Alignment, AutomaticTrimmer
from pytrimal_amd.synth import synth_msa
import cProfile, pstats

a = synth_msa(1000, 4000, 2000)
ali = Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a])
tr = AutomaticTrimmer("automated1", platform="hip")
tr.trim(ali)
ts = []
for _ in range(5):
    t = time.perf_counter(); tr.trim(ali); ts.append(time.perf_counter() - t)
print("trim ms:", [round(x * 1e3, 2) for x in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.trim(ali)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
os.environ["MSA_TRACE"] = "1"
tr.trim(ali)
