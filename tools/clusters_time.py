import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import torch
from pytrimal_amd import Alignment, RepresentativeTrimmer
from pytrimal_amd.synth import synth_msa
for (m, n) in ((1000, 2000), (5000, 5000)):
    a = synth_msa(m, n, 1004)
    ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
    for tr in (RepresentativeTrimmer(clusters=50, platform="hip"), RepresentativeTrimmer(identity_threshold=0.3, platform="hip")):
        tr.trim(ali)
        t = time.perf_counter(); out = tr.trim(ali); dt = time.perf_counter() - t
        print(m, n, repr(tr), "%.1f ms" % (dt * 1e3), "kept", sum(out.sequences_mask), flush=True)
