#!/usr/bin/env python3
"""All BASELINE.json configurations through the public API (host rows -> masks, PCIe-inclusive),
with mask parity against the oracle where the oracle finishes in seconds.  Prints one JSON line
per configuration; the headline contract line is bench.py's."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)

import torch  # noqa: F401,E402  (first HIP runtime in the process, see pytrimal_amd/_lib.py)

import oracle  # noqa: E402
from pytrimal_amd import Alignment, AutomaticTrimmer, ManualTrimmer, RepresentativeTrimmer  # noqa: E402
from pytrimal_amd.synth import synth_msa  # noqa: E402


def ali_of(a):
    return Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a])


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t)
    return out, float(np.median(ts))


def report(name, a, trimmer, oracle_kw, check=True):
    ali = ali_of(a)
    out, sec = timed(lambda: trimmer.trim(ali))
    rec = {"config": name, "m": a.shape[0], "n": a.shape[1], "trimmer": repr(trimmer), "seconds": round(sec, 5),
           "columns_per_s": round(a.shape[1] / sec, 1), "kept_columns": int(sum(out.residues_mask)),
           "kept_sequences": int(sum(out.sequences_mask))}
    if check:
        t = time.perf_counter()
        res, seq, _ = oracle.trim(a, **oracle_kw)
        rec["oracle_seconds"] = round(time.perf_counter() - t, 2)
        rec["masks_equal_oracle"] = bool(out.residues_mask == [bool(x) for x in res] and
                                         out.sequences_mask == [bool(x) for x in seq])
        rec["speedup_vs_oracle_1core"] = round(rec["oracle_seconds"] / sec, 1)
    print(json.dumps(rec), flush=True)


def main():
    which = sys.argv[1:] or ["C2", "C3", "C4", "C5"]
    if "C2" in which:
        report("C2", synth_msa(500, 2000, 1002),
               ManualTrimmer(gap_threshold=0.5, similarity_threshold=0.5, platform="hip"),
               dict(gap_threshold=0.5, similarity_threshold=0.5))
    if "C3" in which:
        report("C3", synth_msa(2000, 10000, 1003), AutomaticTrimmer("automated1", platform="hip"),
               dict(method="automated1"), check="--check-c3" in sys.argv)
    if "C4" in which:
        report("C4", synth_msa(5000, 5000, 1004), RepresentativeTrimmer(identity_threshold=0.5, platform="hip"),
               dict(identity_threshold=0.5), check="--check-c4" in sys.argv)
    if "C5" in which:
        from multiprocessing.pool import ThreadPool

        batch = [synth_msa(1000, 4000, 2000 + k) for k in range(8)]  # one GPU's share of the 64
        alis = [ali_of(a) for a in batch]
        trimmer = AutomaticTrimmer("automated1", platform="hip")
        for threads in (1, 2, 4):
            with ThreadPool(threads) as pool:
                pool.map(trimmer.trim, alis[:threads])
                t = time.perf_counter()
                outs = pool.map(trimmer.trim, alis)
                sec = time.perf_counter() - t
            rec = {"config": "C5 (8 of 64 alignments, one GPU)", "threads": threads, "seconds": round(sec, 4),
                   "columns_per_s": round(8 * 4000 / sec, 1)}
            if threads == 1:
                res, seq, _ = oracle.trim(batch[0], method="automated1")
                rec["masks_equal_oracle_first"] = bool(outs[0].residues_mask == [bool(x) for x in res])
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
