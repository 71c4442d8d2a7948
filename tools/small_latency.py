"""Latency of one trim on small alignments (where launches and synchronisation, not kernels, set the time): public API
and C ABI, by size and method.   python tools/small_latency.py > profiles/rNN_small_latency.jsonl"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch  # noqa: F401
from pytrimal_amd import Alignment, AutomaticTrimmer, ManualTrimmer, _lib
from pytrimal_amd.synth import synth_msa

SIZES = [tuple(int(v) for v in x.split("x")) for x in os.environ["SIZES"].split(",")] if os.environ.get("SIZES") else [(46, 1181), (100, 1000), (200, 2000), (500, 2000), (1000, 4000)]
TRIMMERS = [("gappyout", lambda: AutomaticTrimmer("gappyout", platform="hip")),
            ("strict", lambda: AutomaticTrimmer("strict", platform="hip")),
            ("automated1", lambda: AutomaticTrimmer("automated1", platform="hip")),
            ("manual gap+sim", lambda: ManualTrimmer(gap_threshold=0.5, similarity_threshold=0.5, platform="hip"))]
for m, n in SIZES:
    a = synth_msa(m, n, 77 + m)
    ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
    for name, make in TRIMMERS:
        tr = make()
        for _ in range(5):
            tr.trim(ali)
        N = 100
        best_api = best_abi = 1e9
        names, dense, indet, params, keep = tr._prepare(ali)
        ctx = _lib.thread_context()
        for rep in range(3):
            t = time.perf_counter()
            for _ in range(N):
                tr.trim(ali)
            best_api = min(best_api, (time.perf_counter() - t) / N)
            t = time.perf_counter()
            for _ in range(N):
                ctx.upload(dense, indet)
                ctx.trim(params)
            best_abi = min(best_abi, (time.perf_counter() - t) / N)
        ctx.prof_enable(True)
        ctx.lib.msa_prof_reset(ctx.h)
        for _ in range(10):
            ctx.upload(dense, indet)
            ctx.trim(params)
        kern = {}
        for k in ("front", "gaps", "prep", "pairs", "idstats", "encode", "sim"):
            ms, cnt = ctx.prof_get(k)
            if cnt:
                kern[k] = round(ms / cnt, 4)
        ctx.prof_enable(False)
        print(json.dumps({"m": m, "n": n, "trimmer": name, "public_api_ms": round(best_api * 1e3, 4), "c_abi_ms": round(best_abi * 1e3, 4),
                          "kernels_ms": kern, "kernels_sum_ms": round(sum(kern.values()), 4)}), flush=True)
