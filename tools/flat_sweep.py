"""Where the flat similarity kernel stops paying: one strict trim (upload + msa_trim through the C ABI) of m x n residues with the
flat kernel (MSA_FLAT_MAX_M=512) and with the wave-per-column kernel (MSA_FLAT_MAX_M=0), both inside the compact pipeline.
   python tools/flat_sweep.py [n] > profiles/rNN_flat_sweep.jsonl"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import torch  # noqa: F401
from pytrimal_amd import Alignment, AutomaticTrimmer, _lib
from pytrimal_amd.synth import synth_msa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
for m in (8, 16, 32, 46, 64, 80, 100, 128, 160, 200, 256, 320, 400, 512):
    a = synth_msa(m, n, 77 + m)
    ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
    tr = AutomaticTrimmer("strict", platform="hip")
    rec = {"m": m, "n": n}
    variants = [("flat_ms", "512", None), ("wave_per_column_ms", "0", None)]
    if os.environ.get("FLAT_U_SWEEP"):  # terms per lane and scan of the flat kernel: 4 (rounds 4), 8, 16
        variants = [(f"flat_u{u}_ms", "512", str(u)) for u in (4, 8, 16)] + [("wave_per_column_ms", "0", None)]
    for name, flat, u in variants:
        os.environ["MSA_FLAT_MAX_M"] = flat
        os.environ.pop("MSA_FLAT_U", None)
        if u:
            os.environ["MSA_FLAT_U"] = u
        _lib.reset_thread_context()  # (the library reads the switches when a context is created)
        for _ in range(5):
            tr.trim(ali)
        names, dense, indet, params, keep = tr._prepare(ali)
        ctx = _lib.thread_context()
        best = 1e9
        for rep in range(3):
            t = time.perf_counter()
            for _ in range(300):
                ctx.upload(dense, indet)
                ctx.trim(params)
            best = min(best, (time.perf_counter() - t) / 300)
        ctx.prof_enable(True)
        ctx.lib.msa_prof_reset(ctx.h)
        for _ in range(20):
            ctx.upload(dense, indet)
            ctx.trim(params)
        ms, cnt = ctx.prof_get("sim")
        ctx.prof_enable(False)
        rec[name] = round(best * 1e3, 4)
        rec[name.replace("_ms", "_kernel_ms")] = round(ms / max(cnt, 1), 4)
    print(json.dumps(rec), flush=True)
_lib.reset_thread_context()
