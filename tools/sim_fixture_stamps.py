"""Cycle stamps of the similarity kernel on a REAL alignment (a FASTA file: default the reference's ENOG411BWBU fixture, 209 x 1227)
beside a synthetic one of the same shape: ordered rows per column, cycles by phase, the slowest columns with their residue
make-up.  Conserved columns -- a handful of distinct residues, the numerator a sum of rare large terms -- are what the per-lane
predictor finds hardest.   python tools/sim_fixture_stamps.py [file.fasta]   (MSA_COMPACT=0: the wave-per-column kernel at any size)"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import Alignment, _lib
from pytrimal_amd.synth import synth_msa
from bx_stamps import stamped_similarity

os.environ.setdefault("MSA_COMPACT", "0")
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "data", "ENOG411BWBU.seq40.res60.fasta")
ali = Alignment.load(path, "fasta")
real = np.ascontiguousarray(ali._dense())
m, n = real.shape
lib = _lib.load()
lib.msa_debug_bx_records.argtypes = [ctypes.c_void_p, ctypes.c_int]
for name, a in (("real", real), ("synthetic", synth_msa(m, n, 7))):
    _, _, rec = stamped_similarity(a)
    nw = rec["waves"]
    buf = (ctypes.c_uint * (8 * nw))()
    lib.msa_debug_bx_records(buf, nw)
    r = np.frombuffer(buf, dtype=np.uint32).reshape(nw, 8)
    tot = (r[:, 1].astype(np.int64) + r[:, 2] + r[:, 3]) * 64
    rec.update({"data": name, "m": m, "n": n, "ordered_rows_p50": int(np.percentile(r[:, 6], 50)), "ordered_rows_p90": int(np.percentile(r[:, 6], 90)),
                "ordered_rows_max": int(r[:, 6].max())})
    print(json.dumps(rec), flush=True)
    for i in np.argsort(-tot)[:5]:
        c = int(r[i, 0]); col = a[:, c]
        valid = col[(col != ord("-")) & (col != ord("X"))]
        counts = sorted(np.unique(valid, return_counts=True)[1].tolist(), reverse=True)
        print(json.dumps({"column": c, "kcycles": round(int(tot[i]) / 1e3, 1), "prologue_loops_stitch_kcycles": [round(int(r[i, q]) * 64 / 1e3, 1) for q in (1, 2, 3)], "ordered_rows": int(r[i, 6]), "valid_rows": int(valid.size), "residue_counts": counts[:6]}), flush=True)
