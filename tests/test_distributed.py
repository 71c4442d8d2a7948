"""N > 1 path on CPU: two gloo ranks shard a batch of alignments, trim their shards (with the
oracle standing in for the device, which this container does not have) and gather the masks on
rank 0.  Checks the sharding, the payload packing and the gather against a single-process run.
"""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_batch():
    from pytrimal_amd import Alignment
    from pytrimal_amd.synth import synth_msa

    shapes = [(12, 90), (30, 64), (7, 131), (20, 200), (16, 33)]
    out = []
    for k, (m, n) in enumerate(shapes):
        a = synth_msa(m, n, 500 + k)
        out.append(Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a]))
    return out


def _oracle_trim(ali):
    import oracle
    from pytrimal_amd import TrimmedAlignment

    a = oracle.pack(list(ali.sequences))
    res, seq, _ = oracle.trim(a, method="gappyout")
    return TrimmedAlignment._from_parts(ali.names, a, 0, seq, res)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytrimal_amd import AutomaticTrimmer
        from pytrimal_amd.batch import broadcast_trimmer, shard_indices, trim_batch

        trimmer = AutomaticTrimmer("gappyout") if rank == 0 else AutomaticTrimmer("strict")
        trimmer = broadcast_trimmer(trimmer, src=0)
        assert trimmer.method == "gappyout"
        assert shard_indices(5, world, rank) == list(range(rank, 5, world))
        batch = _make_batch()
        force = world == 1  # (a group of one rank still goes through the gather when asked to: the RCCL one-rank test's path)
        out = trim_batch(trimmer, batch, trim_fn=_oracle_trim, force_collectives=force)
        masks = trim_batch(trimmer, batch, trim_fn=_oracle_trim, masks_only=True, force_collectives=force)  # what the gather moves, nothing rebuilt
        if rank == 0:
            assert len(masks) == len(out)
            for (res, seq), t in zip(masks, out):
                assert res.tolist() == t.residues_mask and seq.tolist() == t.sequences_mask
            q.put([(t.names, list(t.sequences), t.residues_mask, t.sequences_mask) for t in out])
        else:
            assert out is None and masks is None
            q.put(None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 1])
def test_two_rank_batch_matches_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = next(r for r in results if r is not None)
    sys.path.insert(0, ROOT)
    from pytrimal_amd.batch import trim_batch

    want = trim_batch(None, _make_batch(), trim_fn=_oracle_trim)  # no process group: local
    assert len(got) == len(want) == 5
    for g, w in zip(got, want):
        assert g[0] == w.names and g[1] == list(w.sequences)
        assert g[2] == w.residues_mask and g[3] == w.sequences_mask


def test_bench_launcher_spawns_the_ranks():
    """`python bench.py --gpus 2` without a launcher starts two rank processes itself (never touching a GPU in the
    parent); `--launch-check` replaces the GPU workload by a gloo all-reduce so that this runs on a CPU-only host."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2
