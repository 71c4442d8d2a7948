"""BASELINE.json configurations C2-C5 at FULL size through the public API (and `trim_batch`), against the golden
masks of tests/golden/configs.npz (made by tests/golden/make_golden_configs.py with the CPU oracle; the oracle takes
seconds to minutes at these sizes, the HIP path milliseconds).  Mirrors the reference's golden-file tests
(``/root/reference/src/pytrimal/tests/_base.py:15-20``): same inputs, masks compared bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from pytrimal_amd import Alignment, AutomaticTrimmer, ManualTrimmer, RepresentativeTrimmer, _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "configs.npz"))


def ali_of(a):
    return Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a])


def check_masks(golden, key, trimmed, m, n):
    res = np.unpackbits(golden[f"{key}.res"])[:n].astype(bool)
    seq = np.unpackbits(golden[f"{key}.seq"])[:m].astype(bool)
    assert trimmed.residues_mask == res.tolist(), f"{key}: kept-column mask differs from the golden vector"
    assert trimmed.sequences_mask == seq.tolist(), f"{key}: kept-sequence mask differs from the golden vector"


def q_bits(a):
    m = SimilarityMatrix.aa()
    ctx = _lib.Context(0)
    try:
        ctx.upload(a, ord("X"))
        _, q = ctx.similarity(np.ascontiguousarray(m._vhash, dtype=np.int32), np.ascontiguousarray(m._dist, dtype=np.float32))
    finally:
        ctx.close()
    return q.view(np.uint32)


def test_c2_manual_trim_golden(golden):
    a = synth_msa(500, 2000, 1002)
    t = ManualTrimmer(gap_threshold=0.5, similarity_threshold=0.5, platform="hip").trim(ali_of(a))
    check_masks(golden, "C2", t, 500, 2000)
    assert np.array_equal(q_bits(a), golden["C2.q_bits"]), "similarity quotient must be bit-exact"


def test_c3_trim_golden(golden):
    a = synth_msa(2000, 10000, 1003)
    t = AutomaticTrimmer("automated1", platform="hip").trim(ali_of(a))
    check_masks(golden, "C3", t, 2000, 10000)
    assert np.array_equal(q_bits(a), golden["C3.q_bits"]), "similarity quotient must be bit-exact"
    # selectMethod means and the cut points through the C ABI
    ctx = _lib.Context(0)
    m = SimilarityMatrix.aa()
    vhash, dist = np.ascontiguousarray(m._vhash, dtype=np.int32), np.ascontiguousarray(m._dist, dtype=np.float32)
    ctx.upload(a, ord("X"))
    p = _lib.TrimParams(_lib.METHOD_CODES["automated1"], -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0,
                        vhash.ctypes.data, dist.ctypes.data, len(m))
    _, _, info = ctx.trim(p)
    ctx.close()
    assert np.array([info.avg_seq, info.max_seq], dtype=np.float32).view(np.uint32).tolist() == golden["C3.avgmax_bits"].tolist()
    assert [info.selected_method, info.gap_cut] == golden["C3.cuts"].tolist()
    assert np.array([info.sim_cut], dtype=np.float32).view(np.uint32).tolist() == golden["C3.simcut_bits"].tolist()


def test_c3_rank_seeds_golden(golden):
    """The alignments the ranks 1 .. 7 of `bench.py --workload C3 --gpus 8` trim (`seed + rank`: bench.py): masks, selectMethod's
    means and the cut points of `automated1` at the headline's full size for seeds 1004 .. 1010 (round 5's suite knew seed 1003 only)."""
    m, n = 2000, 10000
    mx = SimilarityMatrix.aa()
    vhash, dist = np.ascontiguousarray(mx._vhash, dtype=np.int32), np.ascontiguousarray(mx._dist, dtype=np.float32)
    ctx = _lib.Context(0)
    try:
        p = _lib.TrimParams(_lib.METHOD_CODES["automated1"], -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0,
                            vhash.ctypes.data, dist.ctypes.data, len(mx))
        for seed in range(1004, 1011):
            key = f"C3.s{seed}"
            ctx.upload(synth_msa(m, n, seed), ord("X"))
            res, seq, info = ctx.trim(p)
            assert np.array_equal(res, np.unpackbits(golden[f"{key}.res"])[:n].astype(bool)), f"{key}: kept-column mask"
            assert np.array_equal(seq, np.unpackbits(golden[f"{key}.seq"])[:m].astype(bool)), f"{key}: kept-sequence mask"
            assert np.array([info.avg_seq, info.max_seq], dtype=np.float32).view(np.uint32).tolist() == golden[f"{key}.avgmax_bits"].tolist(), key
            assert [info.selected_method, info.gap_cut] == golden[f"{key}.cuts"].tolist(), key
            assert np.array([info.sim_cut], dtype=np.float32).view(np.uint32).tolist() == golden[f"{key}.simcut_bits"].tolist(), key
    finally:
        ctx.close()


def test_c4_representative_golden(golden):
    a = synth_msa(5000, 5000, 1004)
    t = RepresentativeTrimmer(identity_threshold=0.5, platform="hip").trim(ali_of(a))
    check_masks(golden, "C4", t, 5000, 5000)


def test_c5_batch_golden(golden):
    """The 64 alignments of config 5 through `trim_batch(threads=4)` (one process, no process group: the whole
    batch is this rank's shard) and, for the first eight, one by one: the batch driver must not change results."""
    from pytrimal_amd.batch import trim_batch

    alis = [ali_of(synth_msa(1000, 4000, 2000 + k)) for k in range(64)]
    trimmer = AutomaticTrimmer("automated1", platform="hip")
    out = trim_batch(trimmer, alis, threads=4)
    assert len(out) == 64
    for k, t in enumerate(out):
        check_masks(golden, f"C5.{k}", t, 1000, 4000)
    for k in range(8):
        single = trimmer.trim(alis[k])
        assert single.residues_mask == out[k].residues_mask and single.sequences_mask == out[k].sequences_mask


def test_native_batch_equals_one_by_one():
    """`msa_trim_batch` (native worker threads, one context each) on a mixed bag -- shapes from 3 x 40 to 700 x 900,
    protein and DNA, every trimmer class, an empty alignment, windows -- against `trimmer.trim` one alignment at a
    time and the oracle; then an alignment with a residue outside the similarity matrix (the batch raises what the
    single trim raises); then the per-sequence warnings of a trim that leaves sequences with gaps only."""
    import warnings

    import oracle
    from pytrimal_amd import ManualTrimmer, OverlapTrimmer, RepresentativeTrimmer
    from pytrimal_amd.batch import trim_batch

    rng = np.random.default_rng(5)
    alis, mats = [], []
    for k, (m, n) in enumerate([(3, 40), (64, 256), (700, 900), (65, 129), (200, 2000), (31, 33), (128, 64), (330, 700)]):
        a = synth_msa(m, n, 900 + k)
        if k % 3 == 1:  # DNA
            a = np.frombuffer(b"ACGT-", dtype=np.uint8)[rng.integers(0, 5, (m, n))].copy()
        mats.append(np.ascontiguousarray(a))
        alis.append(Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a]))
    alis.append(Alignment([], []))
    mats.append(np.zeros((0, 0), dtype=np.uint8))
    cases = [(AutomaticTrimmer("automated1", platform="hip"), dict(method="automated1")),
             (AutomaticTrimmer("strictplus", platform="hip"), dict(method="strictplus")),
             (ManualTrimmer(gap_threshold=0.6, similarity_threshold=0.2, window=2, platform="hip"),
              dict(gap_threshold=0.6, similarity_threshold=0.2, window=2)),
             (OverlapTrimmer(50, 0.6, platform="hip"), dict(sequence_overlap=50, residue_overlap=0.6)),
             (RepresentativeTrimmer(identity_threshold=0.4, platform="hip"), dict(identity_threshold=0.4))]
    for trimmer, okw in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = trim_batch(trimmer, alis, threads=5)
            single = [trimmer.trim(x) for x in alis]
            masks = trim_batch(trimmer, alis, threads=5, masks_only=True)
        assert len(out) == len(alis) == len(masks)
        for (res, seq), t in zip(masks, out):
            assert res.tolist() == t.residues_mask and seq.tolist() == t.sequences_mask
        for a, t, s1 in zip(mats, out, single):
            assert t.residues_mask == s1.residues_mask and t.sequences_mask == s1.sequences_mask
            assert t.names == s1.names and list(t.sequences) == list(s1.sequences)
            if a.size:
                res, seq, _ = oracle.trim(a, **okw)
                assert t.residues_mask == [bool(x) for x in res] and t.sequences_mask == [bool(x) for x in seq]
                ends = []
                for x in (t, s1):  # (the batch result counts the gaps on the host, the single trim reuses the trim's vector)
                    try:
                        ends.append(x.terminal_only().residues_mask)
                    except RuntimeError:
                        ends.append(None)
                assert ends[0] == ends[1]
    bad = Alignment([b"a", b"b", b"c"], ["MKKBO", "MKKAY", "MKRAY"])  # 'O' is not in the amino-acid matrix
    with pytest.raises(ValueError):
        AutomaticTrimmer("strict", platform="hip").trim(bad)
    with pytest.raises(ValueError):
        trim_batch(AutomaticTrimmer("strict", platform="hip"), alis[:3] + [bad], threads=3)
    gappy = Alignment([b"x", b"y", b"z", b"w"], ["A--A", "-CC-", "-DD-", "-EE-"])
    for run in (lambda: [ManualTrimmer(gap_threshold=0.6, platform="hip").trim(gappy)],
                lambda: trim_batch(ManualTrimmer(gap_threshold=0.6, platform="hip"), [gappy, alis[0]], threads=2)):
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            t = run()[0]
        assert t.sequences_mask == [False, True, True, True] and t.residues_mask == [False, True, True, False]
        assert [str(w.message) for w in caught if issubclass(w.category, RuntimeWarning)] == ["Removing sequence 'x' composed only by gaps"]


@pytest.mark.parametrize("cols_max", ["", "0"])
def test_batch_engine_against_single_trims(monkeypatch, cols_max):
    """`msa_trim_batch`'s engine (one launch per kernel family for a whole group of alignments, the selection on host-only
    views) on 60 alignments of random shapes up to 500 x 900 with gap-heavy rows and columns, the trimmers it takes (the
    similarity pipeline, the trims that need the gap statistics alone, and -- round 6 -- OverlapTrimmer, RepresentativeTrimmer in both
    modes, noduplicateseqs and gap windows on gap-only trims) --
    against one trim at a time, which goes through an ordinary context.  cols_max: the row count up to which a group's
    similarity statistic runs with a lane per column (default 128; 0: the wave-per-column kernel for every group)."""
    import warnings

    from pytrimal_amd import ManualTrimmer, OverlapTrimmer, RepresentativeTrimmer, _lib
    from pytrimal_amd import batch as batch_mod
    from pytrimal_amd.batch import trim_batch

    if cols_max:
        monkeypatch.setenv("MSA_BATCH_COLS_MAX", cols_max)
    monkeypatch.setenv("MSA_BATCH_ENGINE_MIN", "1")  # (the engine for any number of eligible alignments: by default fewer than 40 go to the workers)
    batch_mod.close_batches()  # (the library reads the switches when the batch object is created)
    rng = np.random.default_rng(77)
    alis = []
    for k in range(60):
        m, n = (int(rng.integers(2, 500)) if k % 2 else int(rng.integers(2, 130))), int(rng.integers(1, 900))
        a = synth_msa(m, n, 3000 + k)
        if k % 5 == 0:  # rows that are nearly all gaps: the trimming may leave them empty (the host-side pass over the rows)
            a[rng.integers(0, m, max(1, m // 10)), :] = ord("-")
            a[0, : max(1, n // 50)] = ord("A")
        if k % 7 == 0:
            a[:, rng.integers(0, n, max(1, n // 4))] = ord("-")
        if k % 6 == 0 and m > 4:  # duplicated rows (noduplicateseqs: the later one stays)
            a[m - 1] = a[1]
            a[m // 2] = a[1]
        alis.append(Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in np.ascontiguousarray(a)]))
    for trimmer in (AutomaticTrimmer("strict", platform="hip"), AutomaticTrimmer("automated1", platform="hip"),
                    AutomaticTrimmer("strictplus", platform="hip"), ManualTrimmer(similarity_threshold=0.3, platform="hip"),
                    AutomaticTrimmer("gappyout", platform="hip"), AutomaticTrimmer("nogaps", platform="hip"),
                    AutomaticTrimmer("noallgaps", platform="hip"), ManualTrimmer(gap_threshold=0.7, platform="hip"),
                    ManualTrimmer(gap_threshold=0.6, similarity_threshold=0.2, conservation_percentage=40, platform="hip"),
                    # (round 6) the trimmers that remove sequences, and a gap window on a gap-only trim: the engine's as well
                    OverlapTrimmer(60.0, 0.5, platform="hip"), OverlapTrimmer(30.0, 0.9, platform="hip"),
                    RepresentativeTrimmer(identity_threshold=0.3, platform="hip"), RepresentativeTrimmer(clusters=3, platform="hip"),
                    AutomaticTrimmer("noduplicateseqs", platform="hip"), ManualTrimmer(gap_threshold=0.7, gap_window=2, platform="hip")):
        with warnings.catch_warnings(record=True) as from_batch:
            warnings.simplefilter("always")
            trim_batch(trimmer, alis, threads=3)
        with warnings.catch_warnings(record=True) as from_batch2:
            warnings.simplefilter("always")
            out = trim_batch(trimmer, alis, threads=3)  # (the second call reuses the arenas: nothing of the first may leak into it)
        with warnings.catch_warnings(record=True) as from_single:
            warnings.simplefilter("always")
            single = [trimmer.trim(x) for x in alis]
        for k, (t, s1) in enumerate(zip(out, single)):
            assert t.residues_mask == s1.residues_mask and t.sequences_mask == s1.sequences_mask, (repr(trimmer), k)
        # the sequences the trimming left with gaps only are reported one by one, in both paths alike
        said = [sorted(str(w.message) for w in ws if issubclass(w.category, RuntimeWarning)) for ws in (from_batch, from_batch2, from_single)]
        assert said[0] == said[1] == said[2]
    # a residue the matrix does not know, in the middle of a batch: the batch raises what the single trim raises
    # (a byte that is not ASCII never gets this far: `Alignment` refuses it)
    weird = Alignment([b"a", b"b", b"c"], ["MKKBO", "MKKAY", "MKRAY"])
    strict = AutomaticTrimmer("strict", platform="hip")
    with pytest.raises(ValueError):
        strict.trim(weird)
    with pytest.raises(ValueError):
        trim_batch(strict, alis[:5] + [weird] + alis[5:9], threads=2)
    batch_mod.close_batches()


def test_batch_engine_return_codes_through_the_c_abi():
    """`msa_trim_batch` on raw matrices (no `Alignment` in front to refuse them): a byte that is not ASCII in one alignment
    of a group gives that alignment MSA_E_NON_ASCII and leaves its neighbours alone -- from the engine's host-only view as
    from an ordinary context."""
    from pytrimal_amd import _lib
    from pytrimal_amd.matrix import SimilarityMatrix

    mat = SimilarityMatrix.aa()
    vhash, dist = mat._device_arrays()
    params = _lib.TrimParams(_lib.METHOD_CODES["strict"], -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0,
                             vhash.ctypes.data, dist.ctypes.data, len(mat))
    mats = [np.ascontiguousarray(synth_msa(30 + 7 * k, 200 + 31 * k, 50 + k)) for k in range(12)]
    mats[4][3, 17] = 0xC3
    os.environ["MSA_BATCH_ENGINE_MIN"] = "1"  # (a dozen alignments would otherwise go to the worker contexts)
    try:
        batch = _lib.Batch(0, 2)
    finally:
        os.environ.pop("MSA_BATCH_ENGINE_MIN")
    try:
        out = batch.trim([(a, ord("X"), params) for a in mats])
    finally:
        batch.close()
    ctx = _lib.Context(0)
    try:
        for k, (a, (res, seq, info, rc, rows)) in enumerate(zip(mats, out)):
            if k == 4:
                assert rc == _lib.E_NON_ASCII
                continue
            assert rc == _lib.OK
            ctx.upload(a, ord("X"))
            keep_res, keep_seq, _ = ctx.trim(params)
            assert np.array_equal(res, keep_res.astype(bool)) and np.array_equal(seq, keep_seq.astype(bool))
        ctx.upload(mats[4], ord("X"))
        with pytest.raises(ValueError):
            ctx.trim(params)
    finally:
        ctx.close()


def test_two_ranks_share_one_gpu():
    """`trim_batch` under a two-rank process group, both ranks on this box's one GPU (gloo carries the gather: RCCL needs
    a GPU per rank): the sharding over ranks, a native batch object per rank and the gather of the masks, against the
    same trims in a single process."""
    import json
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "tests", "measure", "two_rank_batch.py")],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["ranks"] == 2 and rec["alignments"] == 7 and rec["equal_to_single_process"] is True


def test_bench_multi_rank_code_path_on_one_gpu():
    """`python bench.py --gpus 2 --share-gpu`: the launcher, two ranks, BASELINE config 5 sharded over them through the
    native batch path, the gather of the masks, the max-over-ranks timing and the one JSON line -- everything a
    multi-GPU run does except RCCL itself (both ranks use this box's one GPU, the process group runs over gloo)."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--workload", "C5", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["steps"] == 2 and "C5" in rec["config"]["workload"]
    assert rec["config"]["kept_columns"] == 192501  # what the 64 trims keep in total (every mask: test_c5_batch_golden)
    assert rec["strong_scaling_reference_1gpu"]["value"] > 0 and rec["value"] > 0


def test_bench_eight_ranks_dress_rehearsal_on_one_gpu():
    """`python bench.py --gpus 8 --share-gpu --workload C5 --steps 2`: eight ranks on this box's one GPU --
    the launcher, eight ranks x four worker contexts, config 5 sharded 8 x 8, the gather of the masks, the max-over-ranks
    timing, ONE JSON line, a clean exit of every rank -- in well under the time such a run may take.  (RCCL itself needs a
    GPU per rank; the process group runs over gloo and the line says so.)"""
    import json
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--workload", "C5", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=env)
    seconds = time.perf_counter() - t0
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "strong" and rec["steps"] == 2 and "C5" in rec["config"]["workload"]
    assert rec["config"]["ranks_seen"] == 8 and rec["config"]["kept_columns"] == 192501 and rec["config"]["kept_columns_ok"] is True
    assert rec["strong_scaling_reference_1gpu"]["value"] > 0 and rec["value"] > 0
    assert seconds < 180, f"the eight-rank run took {seconds:.0f} s"


def test_bench_default_of_a_scaling_run_on_one_gpu():
    """`python bench.py --gpus 2 --share-gpu --steps 2` with NO --workload: what the driver's scaling run starts for N > 1.  The
    line's `value` is the headline workload (C3, one alignment per rank per step: weak scaling, N x the 1-GPU line's work) and
    BASELINE config 5 -- the batch of 64 sharded over the ranks -- rides in the same line as `c5_batch`, all 64 masks golden."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["steps"] == 2 and "C3" in rec["config"]["workload"]
    assert rec["config"]["ranks_seen"] == 2 and rec["config"]["selected_method"] == "strict" and rec["value"] > 0
    assert rec["config"]["kept_columns"] == int(np.unpackbits(np.load(os.path.join(ROOT, "tests", "golden", "configs.npz"))["C3.res"])[:10000].sum())  # (rank 0's alignment)
    c5 = rec["c5_batch"]
    assert c5["scaling"] == "strong" and c5["n_gpus"] == 2 and c5["kept_columns"] == 192501 and c5["kept_columns_ok"] is True
    assert c5["value"] > 0 and c5["same_batch_on_rank0_alone_ms"] > 0


def _launcher_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def _free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _run_under_launcher(args, timeout):
    """`python -m torch.distributed.run --nproc-per-node 1 ... <args>` on a port that was free a moment ago; another process may
    have taken it since (the socket is closed before the launcher binds it), so a rendezvous that fails to bind is tried again on
    another port instead of surfacing as a test failure."""
    out = None
    for _ in range(3):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                              "--master-port", str(_free_port())] + args, capture_output=True, text=True, timeout=timeout, env=_launcher_env())
        if out.returncode == 0 or not any(w in out.stderr for w in ("Address already in use", "EADDRINUSE", "failed to bind")):
            break
    return out


def test_rccl_one_rank_collectives():
    """RCCL for real, on this box's one GPU: a one-rank `nccl` process group created in a process where libmsastat_hip.so is
    already loaded and has computed (tests/measure/rccl_one_rank.py, a fresh child: the launcher runs before any GPU call) --
    `broadcast_trimmer`, an `all_reduce` of a device tensor, and `trim_batch(..., force_collectives=True)`, whose
    `dist.gather` of the packed uint8 device buffer runs over RCCL even at world size 1; the gathered masks against single
    trims and the oracle, and single trims again with RCCL alive."""
    import json

    out = _run_under_launcher([os.path.join(ROOT, "tests", "measure", "rccl_one_rank.py")], 600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["backend"] == "nccl" and rec["world"] == 1 and rec["all_reduce"] == 1.0 and rec["hip_library_loaded_first"] is True
    # (the broadcast trimmer: the method, and the platform -- spelled out in the repr only where 'hip' is not what "detect" picks)
    assert rec["trimmer_repr"] in ("AutomaticTrimmer('automated1', platform='hip')", "AutomaticTrimmer('automated1')"), rec["trimmer_repr"]
    assert rec["trimmer_platform"] == "hip"
    assert rec["gathered_masks_equal_single"] is True and rec["gathered_objects_equal_single"] is True
    assert rec["trims_after_rccl_equal"] is True and rec["oracle_equal"] is True


def test_bench_one_rank_under_the_launcher_uses_rccl():
    """What the driver's scaling run starts, at N = 1: `python -m torch.distributed.run --nproc-per-node 1 bench.py --workload C5`
    -- the `nccl` backend initialised by bench.py itself, barriers and the max-over-ranks all-reduce on device tensors, the
    gather of the 64 masks over RCCL, ONE JSON line whose `config.backend` says so."""
    import json

    out = _run_under_launcher([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "C5", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], 900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 1 and "C5" in rec["config"]["workload"]
    assert rec["config"]["backend"] == "nccl" and rec["config"]["ranks_seen"] == 1
    assert rec["config"]["kept_columns"] == 192501 and rec["config"]["kept_columns_ok"] is True and rec["value"] > 0


def test_batch_module_in_a_fresh_process():
    """`import pytrimal_amd.batch` + a HIP `trim_batch` in a process that did not import torch first: importing the
    package must not initialise the GPU runtime (the platform is resolved lazily), so that batch's own `import torch`
    still comes before the HIP library is loaded."""
    code = f"""
import sys
sys.path.insert(0, {ROOT!r})
import pytrimal_amd
from pytrimal_amd import _lib
assert _lib._lib is None, 'importing the package loaded the HIP library'
import pytrimal_amd.batch as b
import torch
assert torch.cuda.is_available(), 'torch lost the GPU'
from pytrimal_amd import Alignment, AutomaticTrimmer
from pytrimal_amd.synth import synth_msa
alis = [Alignment([b's%d' % i for i in range(40)], [bytes(r) for r in synth_msa(40, 300, s)]) for s in (1, 2, 3)]
out = b.trim_batch(AutomaticTrimmer('strict', platform='hip'), alis, threads=2)
assert len(out) == 3 and all(len(t.residues_mask) == 300 for t in out)
print('ok')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_batches_from_two_threads_at_once():
    """Two threads calling `trim_batch` on the same device share one native batch object, which takes one call at a
    time: the calls queue up (a refused call used to go unnoticed and return all-ones masks)."""
    import threading

    import oracle
    from pytrimal_amd.batch import trim_batch

    trimmer = AutomaticTrimmer("strict", platform="hip")
    groups = []
    for g in range(2):
        mats = [synth_msa(90 + 30 * k + 7 * g, 400 + 50 * k, 4400 + 10 * g + k) for k in range(6)]
        alis = [Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a]) for a in mats]
        want = [oracle.trim(a, method="strict")[:2] for a in mats]
        groups.append((alis, want))
    failures = []

    def run(alis, want):
        try:
            for _ in range(4):
                out = trim_batch(trimmer, alis, threads=3, masks_only=True)
                for (res, seq), (ores, oseq) in zip(out, want):
                    if not (np.array_equal(res, ores.astype(bool)) and np.array_equal(seq, oseq.astype(bool))):
                        failures.append("masks differ")
        except Exception as err:  # noqa: BLE001
            failures.append(repr(err))

    threads = [threading.Thread(target=run, args=g) for g in groups]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures
