#!/usr/bin/env python3
"""Regenerate tests/golden/configs.npz: the BASELINE.json configurations C2-C5 at FULL size.

Like vectors.npz these are outputs of the CPU oracle (the reference cannot be built or imported here, SURVEY.md
section 0), frozen so that the `-m gpu` tests can compare the HIP path bit for bit at the sizes the benchmark
runs, where the oracle itself takes too long for a test:
  C2  ManualTrimmer(gap_threshold=0.5, similarity_threshold=0.5)   synth_msa(500, 2000, 1002)
  C3  AutomaticTrimmer('automated1')                                synth_msa(2000, 10000, 1003)
  C4  RepresentativeTrimmer(identity_threshold=0.5)                 synth_msa(5000, 5000, 1004)
  C5  AutomaticTrimmer('automated1') x 64                           synth_msa(1000, 4000, 2000 + k), k = 0..63
  C3.s1004 .. C3.s1010 (round 6)  AutomaticTrimmer('automated1')    synth_msa(2000, 10000, 1003 + rank), rank = 1..7: the alignments
      the ranks 1 .. 7 of `bench.py --workload C3 --gpus 8` trim (`seed + rank`)
Stored per configuration: packed kept-column / kept-sequence masks, the selectMethod means (bit patterns), the
cut points, and for C2 / C3 the similarity quotient Q of every column (bit patterns).

Usage:  python tests/golden/make_golden_configs.py        (about two minutes on 8 cores)
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from pytrimal_amd.synth import synth_msa  # noqa: E402


def record(out, key, res, seq, info):
    out[f"{key}.res"] = np.packbits(res)
    out[f"{key}.seq"] = np.packbits(seq)
    out[f"{key}.avgmax_bits"] = np.array([info.avg_seq, info.max_seq], dtype=np.float32).view(np.uint32)
    out[f"{key}.cuts"] = np.array([info.selected, info.gap_cut], dtype=np.int32)
    out[f"{key}.simcut_bits"] = np.array([info.sim_cut], dtype=np.float32).view(np.uint32)


def q_bits(a):
    g, _, _, _ = oracle.gaps(a)
    hit, dst = oracle.pair_counts(a)
    _, q = oracle.similarity(a, oracle.weights(hit, dst), g, *oracle.aa_matrix())
    return q.view(np.uint32)


def c5_one(k):
    res, seq, info = oracle.trim(synth_msa(1000, 4000, 2000 + k), method="automated1")
    avgmax = np.array([info.avg_seq, info.max_seq], dtype=np.float32).view(np.uint32)
    return k, np.packbits(res), np.packbits(seq), avgmax, np.array([info.selected, info.gap_cut], dtype=np.int32)


def c3_rank(seed):
    res, seq, info = oracle.trim(synth_msa(2000, 10000, seed), method="automated1")
    rec = {}
    record(rec, f"C3.s{seed}", res, seq, info)
    return rec


def main():
    out = {}
    a = synth_msa(500, 2000, 1002)
    record(out, "C2", *oracle.trim(a, gap_threshold=0.5, similarity_threshold=0.5))
    out["C2.q_bits"] = q_bits(a)
    a = synth_msa(2000, 10000, 1003)
    record(out, "C3", *oracle.trim(a, method="automated1"))
    out["C3.q_bits"] = q_bits(a)
    a = synth_msa(5000, 5000, 1004)
    record(out, "C4", *oracle.trim(a, identity_threshold=0.5))
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        for k, res, seq, avgmax, cuts in pool.imap_unordered(c5_one, range(64)):
            out[f"C5.{k}.res"], out[f"C5.{k}.seq"], out[f"C5.{k}.avgmax_bits"], out[f"C5.{k}.cuts"] = res, seq, avgmax, cuts
        for rec in pool.imap_unordered(c3_rank, range(1004, 1011)):
            out.update(rec)
    np.savez_compressed(os.path.join(HERE, "configs.npz"), **out)
    print(f"wrote {len(out)} arrays to tests/golden/configs.npz")


if __name__ == "__main__":
    main()
