#!/usr/bin/env python3
"""Regenerate tests/golden/vectors.npz.

The reference (pytrimal + trimAl) cannot be built or imported in this environment (its trimAl
submodule is empty), so these vectors are produced by the CPU oracle
(oracle/msa_oracle.c) AFTER it has been checked against every surviving fixture of the
reference's own test-suite (tests/test_oracle_golden.py, the files under tests/golden/data/).
They freeze the oracle's statistic vectors and masks so that a later change to the oracle or
to the HIP path is caught even where the reference left no fixture.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from conftest import EXAMPLE_001, OVERLAP_EXAMPLE, edge_msa  # noqa: E402
from pytrimal_amd.synth import synth_msa  # noqa: E402


def load(name):
    return oracle.pack(oracle.read_fasta(os.path.join(HERE, "data", name))[1])


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def main():
    out = {}
    inputs = {
        "example001": oracle.pack(EXAMPLE_001),
        "overlapdoc": oracle.pack(OVERLAP_EXAMPLE),
        "enog": load("ENOG411BWBU.seq40.res60.fasta"),
        "pf12574": load("PF12574.full.afa"),
        "halorhodopsin": load("halorhodopsin.afa"),
        "synth64x256": synth_msa(64, 256, 64256),
    }
    wide = oracle.make_matrix((np.arange(23 * 23).reshape(23, 23) % 7 + np.arange(23 * 23).reshape(23, 23).T % 7
                               ).astype(np.float32), "ABCDEFGHIKLMNPQRSTVWXYZ")
    for key, a in inputs.items():
        m, n = a.shape
        g, hist, mx, tot = oracle.gaps(a)
        hit, dst = oracle.pair_counts(a)
        ident = oracle.identities(hit, dst)
        w = oracle.weights(hit, dst)
        out[f"{key}.gaps"] = g
        out[f"{key}.hit_crc"] = crc(hit)
        out[f"{key}.dst_crc"] = crc(dst)
        out[f"{key}.ident_crc"] = crc(ident)
        if m <= 64:
            out[f"{key}.hit"] = hit
            out[f"{key}.dst"] = dst
        sel, avg, mxs = oracle.select_method(ident)
        out[f"{key}.select"] = np.array([sel], dtype=np.int32)
        out[f"{key}.avgmax_bits"] = np.array([avg, mxs], dtype=np.float32).view(np.uint32)
        # lower-case / B / Z residues are outside BLOSUM62's 20 letters: use the wide alphabet there
        matrix = wide if key in ("pf12574",) else oracle.aa_matrix()
        try:
            mdk, q = oracle.similarity(a, w, g, *matrix)
            out[f"{key}.mdk_bits"] = mdk.view(np.uint32)
            out[f"{key}.q_bits"] = q.view(np.uint32)
        except oracle.OracleError as err:
            out[f"{key}.sim_error"] = np.array([err.code, *err.detail], dtype=np.int32)
        for thr in (0.5, 0.6, 0.8):
            out[f"{key}.overlap{int(thr * 100)}_bits"] = oracle.overlap(a, thr).view(np.uint32)
        configs = {
            "gappyout": dict(method="gappyout"), "strict": dict(method="strict"),
            "strictplus": dict(method="strictplus"), "automated1": dict(method="automated1"),
            "nogaps": dict(method="nogaps"), "noallgaps": dict(method="noallgaps"),
            "noduplicateseqs": dict(method="noduplicateseqs"),
            "gt90cons60": dict(gap_threshold=0.9, conservation_percentage=60),
            "gt50st50": dict(gap_threshold=0.5, similarity_threshold=0.5),
            "st30cons50": dict(similarity_threshold=0.3, conservation_percentage=50),
            "gt80w2": dict(gap_threshold=0.8, window=2),
            "ov60_50": dict(sequence_overlap=60, residue_overlap=0.5),
            "id50": dict(identity_threshold=0.5), "clusters3": dict(clusters=3),
        }
        for cname, kw in configs.items():
            if key == "pf12574" and cname in ("strict", "strictplus", "automated1", "gt50st50", "st30cons50"):
                kw = dict(kw, matrix=wide)
            try:
                res, seq, info = oracle.trim(a, **kw)
            except oracle.OracleError as err:
                out[f"{key}.{cname}.error"] = np.array([err.code], dtype=np.int32)
                continue
            out[f"{key}.{cname}.res"] = np.packbits(res)
            out[f"{key}.{cname}.seq"] = np.packbits(seq)
    out["wide.alphabet"] = np.frombuffer(b"ABCDEFGHIKLMNPQRSTVWXYZ", dtype=np.uint8)
    out["wide.dist_bits"] = wide[1].view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **out)
    print(f"wrote {len(out)} arrays to tests/golden/vectors.npz")


if __name__ == "__main__":
    main()
