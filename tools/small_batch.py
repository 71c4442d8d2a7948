"""Many small alignments through trim_batch: `count` x (m x n), strict, ms per batch with the engine (batched kernels) and
with the workers alone (MSA_BATCH_ENGINE=0).   python tools/small_batch.py [count m n [method | overlap | representative]]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import torch  # noqa: F401
from pytrimal_amd import Alignment, AutomaticTrimmer
from pytrimal_amd.batch import trim_batch
from pytrimal_amd.synth import synth_msa

count, m, n = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 100, 1000)))
method = sys.argv[4] if len(sys.argv) > 4 else "strict"
alis = []
for k in range(count):
    a = synth_msa(m, n, 7000 + k)
    alis.append(Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a]))
if method == "overlap":  # (round 6: the trimmers that remove sequences go through the engine as well)
    from pytrimal_amd import OverlapTrimmer
    tr = OverlapTrimmer(60.0, 0.5, platform="hip")
elif method == "representative":
    from pytrimal_amd import RepresentativeTrimmer
    tr = RepresentativeTrimmer(identity_threshold=0.5, platform="hip")
else:
    tr = AutomaticTrimmer(method, platform="hip")
trim_batch(tr, alis, threads=4, masks_only=True)
ts = []
for _ in range(5):
    t = time.perf_counter(); out = trim_batch(tr, alis, threads=4, masks_only=True); ts.append(time.perf_counter() - t)
print(json.dumps({"alignments": count, "m": m, "n": n, "method": method, "engine": os.environ.get("MSA_BATCH_ENGINE", "1"),
                  "ms_best": round(min(ts) * 1e3, 2), "ms_median": round(sorted(ts)[2] * 1e3, 2),
                  "alignments_per_s": round(count / min(ts))}), flush=True)
