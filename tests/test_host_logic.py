"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol of
include/msastat.h, and its selection logic (cut points, column recovery, strict blocks,
clustering) agrees with the oracle and with the reference's fixtures when fed oracle statistics.
No device is needed: these entry points are pure host code.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle
from conftest import EXAMPLE_001, ROOT, data_path, edge_msa
from pytrimal_amd import _lib
from pytrimal_amd.synth import synth_msa


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "msastat.h")).read()
    declared = set(re.findall(r"\b(msa_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.msa_strerror(0) == b"ok"
    assert lib.msa_strerror(_lib.E_WINDOW_TOO_BIG).startswith(b"window")


def test_no_device_means_loud_failure(lib):
    if lib.msa_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = ctypes.c_void_p()
    assert lib.msa_ctx_create(0, ctypes.byref(h)) == _lib.E_NO_DEVICE
    with pytest.raises(RuntimeError):
        _lib.Context(0)


def stats(a):
    g, hist, mx, _ = oracle.gaps(a)
    hit, dst = oracle.pair_counts(a)
    return g, hist, mx, hit, dst


CASES = [("example001", lambda: oracle.pack(EXAMPLE_001)),
         ("enog", lambda: oracle.pack(oracle.read_fasta(data_path("ENOG411BWBU.seq40.res60.fasta"))[1])),
         ("halo", lambda: oracle.pack(oracle.read_fasta(data_path("halorhodopsin.afa"))[1])),
         ("synth", lambda: synth_msa(120, 700, 5))]


@pytest.mark.parametrize("name,make", CASES)
def test_cutpoints_and_gap_cleaning(lib, name, make):
    a = make()
    m, n = a.shape
    g, hist, mx, hit, dst = stats(a)
    assert lib.msa_gaps_cutpoint_2nd_slope(p(g), m, n) == oracle.cutpoint_2nd_slope(hist, m, n, mx)
    for base, thr in [(-1.0, 0.5), (0.0, 0.0), (60.0, float(np.float32(1) - np.float32(0.9))), (40.0, 0.6), (100.0, 0.1)]:
        cut = lib.msa_gaps_cutpoint(p(g), m, n, base, thr)
        assert cut == oracle.gaps_cutpoint(hist, m, n, base, thr)
        for hw in (0, 2):
            if hw > n // 4:
                continue
            gw = np.zeros(n, dtype=np.int32)
            assert lib.msa_window_i32(p(g), n, hw, p(gw)) == 0
            assert np.array_equal(gw, oracle.gaps_window(g, hw))
            keep = np.zeros(n, dtype=np.uint8)
            lib.msa_clean_gaps(p(gw), n, cut, base, p(keep))
            assert np.array_equal(keep.astype(bool), oracle.clean_overpass(gw, cut, base))


@pytest.mark.parametrize("name,make", CASES)
def test_similarity_side_cleaning(lib, name, make):
    a = make()
    m, n = a.shape
    g, hist, mx, hit, dst = stats(a)
    mdk, _ = oracle.similarity(a, oracle.weights(hit, dst), g, *oracle.aa_matrix())
    for hw in (0, 3):
        if hw > n // 4:
            continue
        mw = np.zeros(n, dtype=np.float32)
        assert lib.msa_window_f32(p(mdk), n, hw, p(mw)) == 0
        assert np.array_equal(mw.view(np.uint32), oracle.window_f32(mdk, hw).view(np.uint32))
        for base, thr in [(-1.0, 0.5), (50.0, 0.3), (80.0, 0.9), (0.0, 0.0)]:
            cs = lib.msa_similarity_cutpoint(p(mw), n, base, thr)
            assert cs == oracle.sim_cutpoint(mw, base, thr)
            keep = np.zeros(n, dtype=np.uint8)
            lib.msa_clean_similarity(p(mw), n, cs, base, p(keep))
            assert np.array_equal(keep.astype(bool), oracle.clean_fallbehind(mw, np.float32(cs), base))
            cg = oracle.gaps_cutpoint(hist, m, n, base, 0.4)
            lib.msa_clean_both(p(g), p(mw), n, cg, cs, base, p(keep))
            assert np.array_equal(keep.astype(bool), oracle.clean_both(g, mw, cg, np.float32(cs), base))
        for variable in (0, 1):
            keep = np.zeros(n, dtype=np.uint8)
            gc, sc = ctypes.c_int32(0), ctypes.c_float(0)
            lib.msa_clean_strict(p(g), p(g), p(mw), m, n, variable, p(keep), ctypes.byref(gc), ctypes.byref(sc))
            ogc = oracle.cutpoint_2nd_slope(hist, m, n, mx)
            osc = oracle.comb_simcut(g, mw, ogc)
            assert gc.value == ogc
            assert np.float32(sc.value).view(np.uint32) == osc.view(np.uint32)
            assert np.array_equal(keep.astype(bool), oracle.clean_strict(g, mw, ogc, osc, variable))


@pytest.mark.parametrize("n,seed", [(2047, 1), (2048, 2), (5000, 3), (10000, 4), (20011, 5)])
def test_strict_cleaning_on_long_vectors(lib, n, seed):
    """Synthetic statistic vectors at the lengths where the product's similarity cut selects by histogram instead
    of nth_element (pool >= 2048), with ties, zeros and runs shorter than the minimum block."""
    r = np.random.default_rng(seed)
    m = 400
    g = np.minimum(r.integers(0, m, n) * (r.random(n) < 0.6), m).astype(np.int32)
    hist = np.bincount(g, minlength=m + 1).astype(np.int32)
    mw = np.round(r.random(n) ** 2, 3).astype(np.float32)  # many equal values
    mw[r.random(n) < 0.05] = 0.0
    mx = int(g.max())
    for variable in (0, 1):
        keep = np.zeros(n, dtype=np.uint8)
        gc, sc = ctypes.c_int32(0), ctypes.c_float(0)
        assert lib.msa_clean_strict(p(g), p(g), p(mw), m, n, variable, p(keep), ctypes.byref(gc), ctypes.byref(sc)) == 0
        ogc = oracle.cutpoint_2nd_slope(hist, m, n, mx)
        osc = oracle.comb_simcut(g, mw, ogc)
        assert gc.value == ogc
        assert np.float32(sc.value).view(np.uint32) == osc.view(np.uint32)
        assert np.array_equal(keep.astype(bool), oracle.clean_strict(g, mw, ogc, osc, variable))


def test_window_too_big(lib):
    v = np.zeros(5, dtype=np.int32)
    out = np.zeros(5, dtype=np.int32)
    assert lib.msa_window_i32(p(v), 5, 100, p(out)) == _lib.E_WINDOW_TOO_BIG
    f = np.zeros(5, dtype=np.float32)
    assert lib.msa_window_f32(p(f), 5, 2, p(f.copy())) == _lib.E_WINDOW_TOO_BIG


def test_select_method(lib):
    for avg, mx, m in [(0.6, 0.9, 100), (0.3, 0.5, 100), (0.45, 0.55, 10), (0.45, 0.55, 100), (0.45, 0.7, 100),
                       (0.55, 0.0, 30), (0.38, 0.0, 30)]:
        want = 1 if avg >= 0.55 else 2 if avg <= 0.38 else 1 if m <= 20 else 1 if 0.5 <= mx <= 0.65 else 2
        assert lib.msa_select_method(avg, mx, m) == want


@pytest.mark.parametrize("name,make", CASES[:3])
def test_representatives_match_oracle_and_fixtures(lib, name, make):
    a = make()
    m, n = a.shape
    hit, dst = oracle.pair_counts(a)
    ident = oracle.identities(hit, dst)
    lengths = (a != ord("-")).sum(axis=1).astype(np.int32)
    for thr in (0.2, 0.5, 0.7, 0.75, 0.9):
        keep = np.zeros(m, dtype=np.uint8)
        k = ctypes.c_int32(0)
        lib.msa_representatives(p(ident), p(lengths), m, thr, p(keep), ctypes.byref(k))
        okeep, onc = oracle.representatives(a, ident, np.float32(thr))
        assert k.value == onc and np.array_equal(keep.astype(bool), okeep)
    for clusters in (1, 2, 3, m):
        thr = lib.msa_cutpoint_clusters(p(ident), p(lengths), m, clusters)
        assert np.float32(thr).view(np.uint32) == oracle.cutpoint_clusters(a, ident, clusters).view(np.uint32)
    if name == "enog":  # the reference's own fixture, through the product's host code
        keep = np.zeros(m, dtype=np.uint8)
        lib.msa_representatives(p(ident), p(lengths), m, 0.75, p(keep), None)
        names, seqs = oracle.read_fasta(data_path("ENOG411BWBU.seq40.res60.fasta"))
        en, _ = oracle.read_fasta(data_path("ENOG411BWBU.maxidentity75.fasta"))
        assert [nm for nm, kk in zip(names, keep) if kk] == en


def test_gap_fixture_through_product_host_code(lib):
    # reference fixture cons60.gt90 (tests/test_manual_trimmer.py:32-35) with oracle gap counts
    names, seqs = oracle.read_fasta(data_path("ENOG411BWBU.seq40.res60.fasta"))
    a = oracle.pack(seqs)
    m, n = a.shape
    g = oracle.gaps(a)[0]
    thr = float(np.float32(1) - np.float32(0.9))
    cut = lib.msa_gaps_cutpoint(p(g), m, n, 60.0, thr)
    keep = np.zeros(n, dtype=np.uint8)
    lib.msa_clean_gaps(p(g), n, cut, 60.0, p(keep))
    _, es = oracle.read_fasta(data_path("ENOG411BWBU.cons60.gt90.fasta"))
    assert [bytes(r[keep.astype(bool)]) for r in a] == es


def test_alignment_type_detection_matches_the_oracle():
    """`Alignment._alignment_type` (the host mirror of utils::checkAlignmentType, which picks the default matrix and the
    indetermination symbol) against the oracle's restatement on random alignments -- among them rows whose nucleotide
    share is exactly 0.7f, which upstream compares with the double literal 0.7."""
    from pytrimal_amd.alignment import detect_alignment_type

    aa = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
    nt = np.frombuffer(b"ACGTNRYKMSWBDHVU", dtype=np.uint8)
    rng = np.random.default_rng(5)
    seen = set()
    for it in range(1500):
        m = int(rng.choice([1, 2, 5, 9, 13, 30])) + int(rng.integers(0, 3))
        n = int(rng.choice([10, 20, 33, 100, 250, 700])) + int(rng.integers(0, 7))
        alpha = aa if it % 3 else nt
        a = alpha[rng.integers(0, len(alpha), (m, n))].copy()
        if it % 5 == 0:  # mixtures around the 70 % line
            mix = rng.random((m, n)) < rng.choice([0.6, 0.7, 0.8])
            a = np.where(mix, nt[rng.integers(0, 5, (m, n))], a)
        a[rng.random((m, n)) < rng.beta(0.6, 1.8, n)[None, :]] = ord("-")
        if it % 7 == 0:
            a[rng.random((m, n)) < 0.1] = rng.choice([ord("."), ord("?")])
        a = np.ascontiguousarray(a, dtype=np.uint8)
        t = detect_alignment_type(a)
        assert t == oracle.alignment_type(a), (m, n, it)
        seen.add(t)
    # the exact edge: 14 nucleotide-or-degenerate letters of 20
    row = np.frombuffer(b"ACGTACGTRYKMSW" + b"LLLLLL", dtype=np.uint8)
    edge = np.stack([row, row])
    assert detect_alignment_type(edge) == oracle.alignment_type(edge) == 4
    assert {1, 4} <= seen


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/msastat.h compiles as C99 (`gcc -std=c99 -pedantic -Wall -Werror`) and a C program linked against
    libmsastat_hip.so can call the host-only entry points: the boundary is a C ABI, not a C++ one."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text(r"""
#include <stdio.h>
#include <string.h>
#include "msastat.h"
int main(void) {
    int32_t g[8] = {0, 3, 6, 3, 0, 0, 9, 0}, w[8];
    uint8_t keep[8];
    msa_trim_params p;
    msa_trim_info info;
    memset(&p, 0, sizeof p);
    memset(&info, 0, sizeof info);
    if (msa_window_i32(g, 8, 1, w) != MSA_OK) return 1;
    if (msa_select_method(0.6f, 0.7f, 100) != 1 || msa_select_method(0.2f, 0.3f, 100) != 2) return 2;
    if (msa_clean_gaps(g, 8, 3.0, 0.0f, keep) != MSA_OK) return 3;
    if (strcmp(msa_strerror(MSA_E_WINDOW_TOO_BIG), "window size is too big for this alignment") != 0) return 4;
    if (msa_trim(NULL, &p, keep, keep, &info) != MSA_E_INVALID) return 5;           /* argument checks need no device */
    if (msa_trim_batch(NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) != MSA_E_INVALID) return 6;
    printf("%d %d %d %d %d %d %d %d | %d%d%d%d%d%d%d%d\n", w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], keep[0], keep[1], keep[2],
           keep[3], keep[4], keep[5], keep[6], keep[7]);
    return 0;
}
""")
    exe = tmp_path / "abi"
    libdir = os.path.join(root, "pytrimal_amd")
    cc = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                         "-L", libdir, "-l:libmsastat_hip.so", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0, (run.returncode, run.stderr)
    windowed, mask = run.stdout.strip().split(" | ")
    want = oracle.gaps_window(np.array([0, 3, 6, 3, 0, 0, 9, 0], dtype=np.int32), 1)
    assert [int(x) for x in windowed.split()] == want.tolist()
    assert mask == "".join("1" if x <= 3 else "0" for x in [0, 3, 6, 3, 0, 0, 9, 0])


def test_diagnostic_switches_need_the_master_variable():
    """The MSA_* diagnostic switches are honoured only in a process that sets MSA_DIAGNOSTICS (a stray MSA_SIM_KERNEL=seq in a
    production environment must not cost a factor of ten): `msa_debug_switches_enabled` reads the variable at call time, in a
    fresh process with and without it (this suite sets it in conftest.py)."""
    import subprocess
    import sys

    code = ("import ctypes, sys; L = ctypes.CDLL(sys.argv[1]); L.msa_debug_switches_enabled.restype = ctypes.c_int; "
            "print(L.msa_debug_switches_enabled())")
    from pytrimal_amd import _lib

    for env_extra, want in ((dict(MSA_DIAGNOSTICS="1"), "1"), (dict(), "0")):
        env = {k: v for k, v in os.environ.items() if not k.startswith("MSA_")}
        env.update(env_extra)
        env["MSA_SIM_KERNEL"] = "seq"  # (never enough by itself)
        out = subprocess.run([sys.executable, "-c", code, _lib.LIB_PATH], capture_output=True, text=True, env=env, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.strip() == want
