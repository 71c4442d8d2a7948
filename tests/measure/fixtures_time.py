"""The reference's own test alignments (tests/golden/data) through every trimmer: public-API latency on the GPU beside the
CPU oracle's time for the same trim (one core), masks compared.   python tests/measure/fixtures_time.py > profiles/rNN_fixtures_time.jsonl"""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import oracle
from pytrimal_amd import Alignment, AutomaticTrimmer, ManualTrimmer, OverlapTrimmer, RepresentativeTrimmer

DATA = os.path.join(ROOT, "tests", "golden", "data")
FIXTURES = ["ENOG411BWBU.seq40.res60.fasta", "halorhodopsin.afa", "PF12574.full.afa"]
CASES = [("AutomaticTrimmer('strictplus')", lambda: AutomaticTrimmer("strictplus", platform="hip"), dict(method="strictplus")),
         ("AutomaticTrimmer('automated1')", lambda: AutomaticTrimmer("automated1", platform="hip"), dict(method="automated1")),
         ("AutomaticTrimmer('gappyout')", lambda: AutomaticTrimmer("gappyout", platform="hip"), dict(method="gappyout")),
         ("ManualTrimmer(gap_threshold=0.9, conservation_percentage=60)", lambda: ManualTrimmer(gap_threshold=0.9, conservation_percentage=60, platform="hip"),
          dict(gap_threshold=0.9, conservation_percentage=60)),
         ("ManualTrimmer(similarity_threshold=0.5)", lambda: ManualTrimmer(similarity_threshold=0.5, platform="hip"), dict(similarity_threshold=0.5)),
         ("OverlapTrimmer(80, 0.8)", lambda: OverlapTrimmer(80.0, 0.8, platform="hip"), dict(sequence_overlap=80.0, residue_overlap=0.8)),
         ("RepresentativeTrimmer(identity_threshold=0.75)", lambda: RepresentativeTrimmer(identity_threshold=0.75, platform="hip"), dict(identity_threshold=0.75))]
for f in FIXTURES:
    ali = Alignment.load(os.path.join(DATA, f), "fasta")
    a = oracle.pack(list(ali.sequences))
    for name, make, kw in CASES:
        tr = make()
        try:
            out = tr.trim(ali)
        except Exception as e:  # the oracle must refuse it as well
            try:
                oracle.trim(a, **kw)
                same = False
            except Exception:
                same = True
            print(json.dumps({"fixture": f, "shape": list(a.shape), "trimmer": name, "raises": type(e).__name__, "oracle_raises_too": same}), flush=True)
            continue
        for _ in range(3):
            tr.trim(ali)
        ts = []
        for _ in range(30):
            t = time.perf_counter()
            tr.trim(ali)
            ts.append(time.perf_counter() - t)
        t = time.perf_counter()
        res, seq, _ = oracle.trim(a, **kw)
        cpu = time.perf_counter() - t
        same = out.residues_mask == [bool(x) for x in res] and out.sequences_mask == [bool(x) for x in seq]
        print(json.dumps({"fixture": f, "shape": list(a.shape), "trimmer": name, "gpu_public_api_ms_median": round(statistics.median(ts) * 1e3, 4),
                          "gpu_public_api_ms_min": round(min(ts) * 1e3, 4), "cpu_oracle_one_core_ms": round(cpu * 1e3, 3),
                          "kept_columns": int(sum(out.residues_mask)), "kept_sequences": int(sum(out.sequences_mask)), "masks_equal": bool(same)}), flush=True)
