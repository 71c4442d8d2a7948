"""The C5 batch (64 x 1000 x 4000, automated1) through msa_trim_batch's engine: ms per batch by the number of groups the call
is cut into (four; round 4 measured 1 / 4 / 16 through a switch that is gone: profiles/r04_c5_engine.jsonl), engine off for comparison.   python tools/c5_engine.py [count]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import torch  # noqa: F401
from pytrimal_amd import Alignment, AutomaticTrimmer
from pytrimal_amd.batch import trim_batch
from pytrimal_amd.synth import synth_msa

count = int(sys.argv[1]) if len(sys.argv) > 1 else 64
alis = []
for k in range(count):
    a = synth_msa(1000, 4000, 2000 + k)
    alis.append(Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a]))
tr = AutomaticTrimmer("automated1", platform="hip")
trim_batch(tr, alis, threads=4)
ts = []
for _ in range(7):
    t = time.perf_counter(); out = trim_batch(tr, alis, threads=4); ts.append(time.perf_counter() - t)
print(json.dumps({"alignments": count, "engine": os.environ.get("MSA_BATCH_ENGINE", "1"), 
                  "ms_best": round(min(ts) * 1e3, 2), "ms_median": round(sorted(ts)[3] * 1e3, 2),
                  "kept_columns": int(sum(sum(t.residues_mask) for t in out))}), flush=True)
