"""The reference's own benchmark shape (README.md:156-160: 3583 sequences x 7287 columns, protein) on synthetic
data: the four statistic -> trimmer mappings of bench/bench.py:48-57, whole trim() calls."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import numpy as np
from pytrimal_amd import Alignment, ManualTrimmer, OverlapTrimmer, RepresentativeTrimmer, _lib
from pytrimal_amd.synth import synth_msa

m, n = 3583, 7287
a = synth_msa(m, n, 3583)
ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
published = {"Gaps": 0.0052, "Similarity": 7.34, "Overlap": 7.68, "Identity": 8.57}  # AVX2, i7-10710U 1 core
trimmers = {"Gaps": ManualTrimmer(gap_threshold=0.5, platform="hip"),
            "Similarity": ManualTrimmer(similarity_threshold=0.5, platform="hip"),
            "Overlap": OverlapTrimmer(60.0, 0.5, platform="hip"),
            "Identity": RepresentativeTrimmer(identity_threshold=0.5, platform="hip")}
for name, tr in trimmers.items():
    tr.trim(ali)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); out = tr.trim(ali); ts.append(time.perf_counter() - t)
    sec = float(np.median(ts))
    ctx = _lib.thread_context()
    ctx.prof_enable(True); ctx.prof_reset(); tr.trim(ali)
    k = {nm: round(ctx.prof_get(nm)[0] / max(1, ctx.prof_get(nm)[1]), 3) for nm in ("pairs", "sim", "simnum", "simden", "overlap", "gaps", "cluster") if ctx.prof_get(nm)[1]}
    ctx.prof_enable(False)
    print(json.dumps({"statistic": name, "trimmer": repr(tr), "m": m, "n": n, "seconds": round(sec, 5),
                      "columns_per_s": round(n / sec), "reference_avx2_seconds_i7_10710U": published[name],
                      "ratio": round(published[name] / sec, 1), "kernels_ms": k,
                      "kept": [int(sum(out.residues_mask)), int(sum(out.sequences_mask))]}), flush=True)
