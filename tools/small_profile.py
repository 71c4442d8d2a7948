"""Per-call latency of small trims (C1 = ENOG 209 x 1227 strictplus, C2 = 500 x 2000 manual)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import cProfile, pstats
from pytrimal_amd import Alignment, AutomaticTrimmer, ManualTrimmer
from pytrimal_amd.synth import synth_msa

cases = [("C1 ENOG strictplus", Alignment.load(os.path.join(ROOT, "tests/golden/data/ENOG411BWBU.seq40.res60.fasta")), AutomaticTrimmer("strictplus", platform="hip"))]
a = synth_msa(500, 2000, 1002)
cases.append(("C2 manual", Alignment([b"s%d" % i for i in range(500)], [bytes(r) for r in a]), ManualTrimmer(gap_threshold=0.5, similarity_threshold=0.5, platform="hip")))
for name, ali, tr in cases:
    tr.trim(ali)
    ts = []
    for _ in range(20):
        t = time.perf_counter(); tr.trim(ali); ts.append(time.perf_counter() - t)
    ts.sort()
    print(name, "trim ms: median %.3f min %.3f" % (ts[10] * 1e3, ts[0] * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): tr.trim(ali)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
    os.environ["MSA_TRACE"] = "1"; tr.trim(ali); os.environ.pop("MSA_TRACE")
