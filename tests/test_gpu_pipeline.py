"""msa_trim's similarity pipeline (everything enqueued before the first wait, automated1 gated on the device) against
the oracle and against the same library with MSA_PIPELINE=0 (the serial flow: wait for the gap counts, wait for the
identity statistics, then enqueue the similarity pass)."""
import ctypes

import numpy as np
import pytest

import oracle
from pytrimal_amd import AutomaticTrimmer, ManualTrimmer, _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

pytestmark = pytest.mark.gpu
ALPHA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)


@pytest.fixture
def contexts(monkeypatch):
    """(pipelined, serial, pipelined with the side stream at any size, row-index lists at any size, and the three other
    ways a small alignment can go) contexts: the switches are read when a context is created"""
    made = []
    for value in ("1", "0", "3"):
        monkeypatch.setenv("MSA_PIPELINE", value)
        made.append(_lib.Context(0))
    monkeypatch.delenv("MSA_PIPELINE")
    monkeypatch.setenv("MSA_LG_BIG", "1")  # (and one with the similarity kernel's row-index lists at any size)
    made.append(_lib.Context(0))
    monkeypatch.delenv("MSA_LG_BIG")
    # small alignments (the first context runs them through the compact pipeline -- three launches, the flat similarity kernel up
    # to 128 sequences, rows of less than 96 KB read in place): the ordinary launch sequence with the rows copied; the compact
    # pipeline with the flat kernel wherever it applies; the compact pipeline with the wave-per-column kernel
    for env in (dict(MSA_COMPACT="0", MSA_ZEROCOPY_KB="0"), dict(MSA_FLAT_MAX_M="512"), dict(MSA_FLAT_MAX_M="0")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        made.append(_lib.Context(0))
        for k in env:
            monkeypatch.delenv(k)
    yield made
    for c in made:
        c.close()


def params_of(trimmer):
    mx = SimilarityMatrix.aa()
    vhash = np.ascontiguousarray(mx._vhash, dtype=np.int32)
    dist = np.ascontiguousarray(mx._dist, dtype=np.float32)
    p = _lib.TrimParams(0, -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, None, None, 0)
    trimmer._configure(p)
    p.vhash = vhash.ctypes.data_as(ctypes.c_void_p)
    p.dist = dist.ctypes.data_as(ctypes.c_void_p)
    p.npos = len(mx)
    return p, (vhash, dist)  # (the arrays must outlive the parameter block)


def family(m, n, seed, keep):
    """m copies of a random root, each residue kept with probability `keep` (else redrawn), 8 % gaps"""
    r = np.random.default_rng(seed)
    root = ALPHA[r.integers(0, 20, n)]
    a = np.where(r.random((m, n)) < keep, root[None, :], ALPHA[r.integers(0, 20, (m, n))])
    a[r.random((m, n)) < 0.08 * r.random(n)[None, :] * 4] = ord("-")
    return np.ascontiguousarray(a, dtype=np.uint8)


def run_both(contexts, a, trimmer, **oracle_kw):
    p, keepalive = params_of(trimmer)
    res, seq, oinfo = oracle.trim(a, **oracle_kw)
    infos = []
    for ctx in contexts:
        for _ in range(2):  # (twice: the second call runs over the buffers the first one left)
            ctx.upload(a, ord("X"))
            keep_res, keep_seq, info = ctx.trim(p)
            assert np.array_equal(keep_res, res)
            assert np.array_equal(keep_seq, seq)
        infos.append(info)
    del keepalive
    return oinfo, infos


CASES = {  # name: (m, n, seed, keep) -> what Cleaner::selectMethod makes of it
    "conserved": (60, 300, 1, 0.92),   # mean identity >= 0.55: gappyout
    "diverged": (60, 300, 2, 0.30),    # mean identity <= 0.38: strict
    "few": (12, 200, 3, 0.70),         # in between, <= 20 sequences: gappyout
    "middle": (80, 260, 4, 0.72),      # in between, mean of the row maxima below 0.5: strict
    "middle_max": (80, 260, 7, 0.85),  # in between, mean of the row maxima in [0.5, 0.65]: gappyout
    "large": (700, 900, 6, 0.45),
    # (from 513 sequences on the default context's compact pipeline sorts its columns behind the front kernel's event: the gate
    # up -- gappyout, the similarity kernel's waves leave at once -- and down)
    "large_conserved": (600, 700, 8, 0.93),
    "wide": (150, 5300, 9, 0.40),      # more columns than the chip has wave slots: sorted as well
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_automated1_gate(contexts, case):
    m, n, seed, keep = CASES[case]
    a = family(m, n, seed, keep)
    oinfo, infos = run_both(contexts, a, AutomaticTrimmer("automated1", platform="hip"), method="automated1")
    for info in infos:
        assert info.selected_method == oinfo.selected
        assert np.float32(info.avg_seq).view(np.uint32) == np.float32(oinfo.avg_seq).view(np.uint32)
        assert np.float32(info.max_seq).view(np.uint32) == np.float32(oinfo.max_seq).view(np.uint32)
        assert info.gap_cut == oinfo.gap_cut
        if oinfo.selected == 2:
            assert np.float32(info.sim_cut).view(np.uint32) == np.float32(oinfo.sim_cut).view(np.uint32)


def test_automated1_cases_cover_both_methods():
    picked = {name: oracle.trim(family(*CASES[name]), method="automated1")[2].selected for name in CASES}
    assert set(picked.values()) == {1, 2}, picked
    assert picked == {"conserved": 1, "diverged": 2, "few": 1, "middle": 2, "middle_max": 1, "large": 2, "large_conserved": 1, "wide": 2}


@pytest.mark.parametrize("kw", [dict(method="strict"), dict(method="strictplus"),
                                dict(similarity_threshold=0.4), dict(gap_threshold=0.6, similarity_threshold=0.3),
                                dict(gap_threshold=0.7, similarity_threshold=0.2, gap_window=2, similarity_window=3),
                                dict(similarity_threshold=0.3, window=4, conservation_percentage=40)])
@pytest.mark.parametrize("shape", [(9, 70, 0.5), (130, 333, 0.4), (513, 1100, 0.5)])
def test_pipelined_methods(contexts, kw, shape):
    m, n, keep = shape
    a = family(m, n, 11 + m, keep)
    trimmer = AutomaticTrimmer(kw["method"], platform="hip") if "method" in kw else ManualTrimmer(platform="hip", **kw)
    run_both(contexts, a, trimmer, **kw)


@pytest.mark.parametrize("method", ["strict", "automated1", "gappyout"])
def test_every_path_reports_the_same_about_a_trim(contexts, method):
    """What a trim says beside its masks -- the warnings (two sequences of gaps only: no identity between them is defined, and
    the trimming drops them), the row it names, both cut points and selectMethod's two statistics -- from every way a small
    alignment can go (the compact pipeline with either similarity kernel, the ordinary launch sequence, the serial flow)."""
    for m, n in ((40, 300), (150, 700)):
        a = family(m, n, 90 + m, 0.5).copy()
        a[3, :] = ord("-")
        a[m - 2, :] = ord("-")
        p, keepalive = params_of(AutomaticTrimmer(method, platform="hip"))
        seen = []
        for ctx in contexts:
            ctx.upload(a, ord("X"))
            keep_res, keep_seq, info = ctx.trim(p)
            seen.append((keep_res.tobytes(), keep_seq.tobytes(), info.warnings, info.warn_row, info.gap_cut,
                         np.float32(info.sim_cut).view(np.uint32).item(), np.float32(info.avg_seq).view(np.uint32).item(),
                         np.float32(info.max_seq).view(np.uint32).item(), info.selected_method, info.kept_residues, info.kept_sequences))
        assert all(x == seen[0] for x in seen), (method, m, n)
        assert seen[0][2] & _lib.W_ONLY_GAPS_SEQUENCES and seen[0][3] == 3
        if method != "gappyout":
            assert seen[0][2] & _lib.W_UNDEFINED_IDENTITY
        res, seq, _ = oracle.trim(a, method=method)
        assert seen[0][0] == np.asarray(res, dtype=np.uint8).tobytes() and seen[0][1] == np.asarray(seq, dtype=np.uint8).tobytes()
        del keepalive


def test_shapes_alternate_on_one_context(contexts):
    """the padding of the float matrices is zeroed per shape, not per pass: a context that sees a large alignment,
    a small one and the large one again must not read what the other shape left behind"""
    big, small, odd = synth_msa(300, 500, 71), synth_msa(70, 900, 72), synth_msa(257, 130, 73)
    p, keepalive = params_of(AutomaticTrimmer("automated1", platform="hip"))
    ps, keepalive2 = params_of(AutomaticTrimmer("strict", platform="hip"))
    expected = {id(x): oracle.trim(x, method="automated1")[0] for x in (big, small, odd)}
    expected_s = {id(x): oracle.trim(x, method="strict")[0] for x in (big, small, odd)}
    for ctx in contexts:
        for a in (big, small, big, odd, small, odd, big):
            ctx.upload(a, ord("X"))
            assert np.array_equal(ctx.trim(p)[0], expected[id(a)])
            ctx.upload(a, ord("X"))
            assert np.array_equal(ctx.trim(ps)[0], expected_s[id(a)])
    del keepalive, keepalive2


def test_trims_repeated_on_one_upload(contexts):
    """several trims of ONE upload (cached planes, matrices and gap counts; the gate word of an earlier automated1 still
    set when a method that always needs the similarity values follows)"""
    for name in ("conserved", "diverged"):
        a = family(*CASES[name])
        plan = [("automated1", AutomaticTrimmer("automated1", platform="hip")), ("strict", AutomaticTrimmer("strict", platform="hip")),
                ("automated1", AutomaticTrimmer("automated1", platform="hip")), ("gappyout", AutomaticTrimmer("gappyout", platform="hip")),
                ("strictplus", AutomaticTrimmer("strictplus", platform="hip"))]
        expected = {method: oracle.trim(a, method=method)[0] for method, _ in plan}
        for ctx in contexts:
            ctx.upload(a, ord("X"))
            for method, trimmer in plan:
                p, keepalive = params_of(trimmer)
                assert np.array_equal(ctx.trim(p)[0], expected[method]), (name, method)
                del keepalive


def test_bad_residue_only_matters_when_similarity_is_used(contexts):
    """automated1 encodes the columns for the similarity pass before it knows whether strict will be selected: a
    symbol outside the matrix must raise exactly when the reference would have reached the similarity statistic"""
    for name in ("conserved", "diverged", "large", "large_conserved"):
        a = family(*CASES[name]).copy()
        a[5, 17] = ord("J")  # not in the default matrix
        p, keepalive = params_of(AutomaticTrimmer("automated1", platform="hip"))
        try:
            res = oracle.trim(a, method="automated1")[0]
        except oracle.OracleError:
            res = None
        for ctx in contexts:
            ctx.upload(a, ord("X"))
            if res is None:
                with pytest.raises(ValueError):
                    ctx.trim(p)
            else:
                assert np.array_equal(ctx.trim(p)[0], res)
            # the context stays usable and clean after the error
            b = family(*CASES["diverged"])
            ctx.upload(b, ord("X"))
            assert np.array_equal(ctx.trim(p)[0], oracle.trim(b, method="automated1")[0])
        del keepalive


def test_repeated_similarity_on_one_alignment(contexts):
    """two similarity passes over one upload (the first-bad-residue key is reset in between)"""
    a = family(90, 150, 5, 0.5)
    vhash, dist = oracle.aa_matrix()
    hit, dst = oracle.pair_counts(a, ord("X"))
    mdk0, _ = oracle.similarity(a, oracle.weights(hit, dst), None, vhash, dist)
    for ctx in contexts:
        ctx.upload(a, ord("X"))
        for k in range(3):
            mdk, _ = ctx.similarity(vhash, dist)
            assert np.array_equal(np.asarray(mdk, dtype=np.float32).view(np.uint32), np.asarray(mdk0, dtype=np.float32).view(np.uint32))
            if k == 0:  # a pass that fails in between (a matrix without 'A') must not leave its key behind
                holed = np.array(vhash, dtype=np.int32).copy()
                holed[0] = -1
                with pytest.raises(ValueError):
                    ctx.similarity(holed, dist)


@pytest.mark.parametrize("seed", [31, 32])
def test_random_trims_against_the_oracle(seed):
    """tests/fuzz/fuzz_trim.py: random shapes, compositions and trimmer settings through msa_trim under seven switch settings
    (default, serial flow, side stream at any size, dense pair codes at any size, the raw pair loops) against the oracle's
    trim: masks, the selected method and the identity mean, and errors raised exactly where the oracle raises."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_trim.py"), "8", str(seed)], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["mismatch"] is False and line["cases"] > 100


def test_random_tall_trims_against_the_oracle():
    """tests/fuzz/fuzz_trim.py in its `tall` mode (round 6): 700 ... 2300 sequences, dense around 1024 / 1025, 1799 / 1800 and
    512 / 513 -- the compact pipeline with its columns dealt by weight, the hand-over to the ordinary pipeline, six rounds per launch,
    the narrow front kernel and the sixteen-row pair tiles, i.e. the paths the BASELINE's C3 and C5 take -- on random data, every
    trimmer kind, fifteen contexts per case, the oracle per case."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_trim.py"), "10", "33", "tall"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["mismatch"] is False and line["mode"] == "tall" and line["cases"] > 30, line


@pytest.mark.parametrize("m,n,kernel", [(8990, 20, "lg_pipe"), (11050, 20, "lg_xseg")])
def test_whole_trims_on_both_sides_of_the_tall_boundary(m, n, kernel):
    """Up to 9000 rows (11000 with a compute unit per column) a split column's loop waves run ahead of its service wave inside one workgroup; beyond, wave w of every column
    runs on XCD w and the increments travel through memory (round 6, late).  Whole `strict` and `automated1` trims of an alignment on
    either side, through the DEFAULT dispatch, against the oracle: masks, cuts, the selected method."""
    rng = np.random.default_rng(m)
    a = synth_msa(m, n, 5150 + m)
    a[rng.random(a.shape) < 0.02] = ord("X")
    a[: m // 3, n // 2] = ord("-")  # (a column whose first valid row lies a hundred rounds down)
    matrix = SimilarityMatrix.aa()
    vhash = np.ascontiguousarray(matrix._vhash, dtype=np.int32)
    dist = np.ascontiguousarray(matrix._dist, dtype=np.float32)
    ctx = _lib.Context(0)
    try:
        for method in ("strict", "automated1"):
            P = _lib.TrimParams(_lib.METHOD_CODES[method], -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, vhash.ctypes.data, dist.ctypes.data,
                                len(matrix))
            ctx.upload(a, ord("X"))
            res, seq, info = ctx.trim(P)
            if method == "strict":
                assert ctx.last_paths()["sim_kernel"] == kernel, ctx.last_paths()
            ores, oseq, oinfo = oracle.trim(a, method=method)
            assert np.array_equal(res, ores) and np.array_equal(seq, oseq), f"{method}: masks differ from the oracle"
            assert info.gap_cut == oinfo.gap_cut and info.selected_method == oinfo.selected
            assert np.float32(info.sim_cut).view(np.uint32) == np.float32(oinfo.sim_cut).view(np.uint32)
    finally:
        ctx.close()


def test_public_api_from_threads_against_the_oracle():
    """tests/fuzz/fuzz_threads.py: four threads trimming random protein / DNA / RNA alignments through the four trimmer classes
    (type detection, default matrices, per-thread contexts) against the oracle's trim at the same time."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_threads.py"), "10", "4"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["mismatch"] is False and min(line["trims_per_thread"]) > 20, line


@pytest.mark.parametrize("seed", [41, 42])
def test_random_batches_against_the_oracle(seed):
    """tests/fuzz/fuzz_batch.py: random batches of 2 .. 300 alignments with a trimmer setting each through `msa_trim_batch` (the
    batched-kernel engine for the small alignments whose trim it takes, worker contexts for the others) against the oracle's
    trim, alignment by alignment: masks and return codes."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_batch.py"), "10", str(seed)], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["mismatch"] is False and line["alignments"] > 50, line
