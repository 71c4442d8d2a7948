"""Batches of independent alignments across the GPUs of a node.

Alignments are independent units (the reference treats them so: one local `trimAlManager` per
`trim` call, ``/root/reference/src/pytrimal/_trimal.pyx:1314-1316``, and a thread pool over
alignments in ``README.md:136-152``), so the path shards at alignment granularity: one process per
GPU, static round-robin of the alignments over the ranks, no collective inside an alignment.
The only communication is control-plane sized: an optional broadcast of the trimmer from rank 0
and a gather of the kept-column / kept-sequence masks to rank 0 (`torch.distributed`, backend
"nccl" = RCCL on ROCm for device tensors, "gloo" for the CPU tests).

Import order matters in a process that uses PyTorch-ROCm: this module imports torch before the
HIP library is loaded (see pytrimal_amd._lib).
"""
import atexit
import os
import threading

import numpy as np
import torch
import torch.distributed as dist

from .trimmer import _raise_warnings
from .alignment import Alignment, TrimmedAlignment

# One native batch object (worker threads + their device contexts: O(m^2) buffers each) per (device, workers), kept
# between calls, closed at exit, rebuilt in a child process (worker threads do not survive a fork).
_BATCHES = {}
_BATCHES_PID = None
_BATCHES_LOCK = threading.Lock()


def _close_batches():
    for b in list(_BATCHES.values()):
        try:
            b.close()
        except Exception:
            pass
    _BATCHES.clear()


atexit.register(_close_batches)


def close_batches():
    """Close this process's native batch objects (worker threads, device contexts and arenas); the next `trim_batch` creates
    new ones, which read the MSA_BATCH_* / MSA_* diagnostic switches again."""
    with _BATCHES_LOCK:
        _close_batches()


def _native_batch(device_index, workers):
    from . import _lib

    global _BATCHES_PID
    with _BATCHES_LOCK:
        if _BATCHES_PID != os.getpid():
            _BATCHES.clear()  # (inherited across a fork: the threads are gone and the handles belong to the parent)
            _BATCHES_PID = os.getpid()
        key = (device_index, workers)
        b = _BATCHES.get(key)
        if b is None:
            for other in [k for k in _BATCHES if k[0] == device_index]:  # one set of contexts per device
                _BATCHES.pop(other).close()
            b = _BATCHES[key] = _lib.Batch(device_index, workers)
        return b


def shard_indices(n_items, world_size, rank):
    """Static round-robin: item i belongs to rank i % world_size."""
    return list(range(rank, n_items, world_size))


def broadcast_trimmer(trimmer, src=0, group=None):
    """Make every rank use rank `src`'s trimmer (pickle state, a few hundred bytes)."""
    if not (dist.is_available() and dist.is_initialized()):
        return trimmer
    box = [trimmer if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def _pack_masks(results):
    """[(keep_res bool[n], keep_seq bool[m]), ...] -> one uint8 vector."""
    parts = []
    for res, seq in results:
        parts.append(np.asarray(res, dtype=np.uint8))
        parts.append(np.asarray(seq, dtype=np.uint8))
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)


def trim_batch(trimmer, alignments, matrix=None, *, group=None, device=None, trim_fn=None, threads=6, shard=True,
               masks_only=False, force_collectives=False):
    """Trim `alignments` (the same list on every rank) with `trimmer`, sharded over the ranks of
    `group`.  Returns the list of `TrimmedAlignment` on rank 0 and `None` elsewhere; without an
    initialised process group it simply trims everything locally.

    Within a rank the shard goes to the native batch path (`msa_trim_batch`, include/msastat.h): `threads` worker
    threads inside the library, each with its own device context, take the alignments largest first; uploads, kernels
    and host selection logic of different alignments overlap on the GPU and the interpreter lock is released for
    the whole shard (one 1000 x 4000 alignment does not fill the chip, and the host side of a trim is serial).

    `trim_fn(alignment) -> TrimmedAlignment` replaces the device path (used by the CPU tests,
    which have no device): the shard is then trimmed one alignment after the other in the calling thread.
    `shard=False`: ignore the process group and trim the whole list on this rank (returns the list).
    `masks_only=True`: return `(residues_mask, sequences_mask)` pairs of bool arrays instead of `TrimmedAlignment`
    objects -- what the gather moves; building 64 result objects is interpreter time behind the device's work (and, on
    rank 0 of a sharded run, serial work for every other rank's alignments).
    `force_collectives=True`: run the gather of the masks even in a process group of ONE rank (where it moves nothing): the
    collective path of a multi-GPU run -- device buffer, `dist.gather` over RCCL, unpacking -- on a one-GPU box
    (tests/measure/rccl_one_rank.py, `bench.py` under a launcher).
    """
    _mark("start")
    distributed = shard and dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    mine = shard_indices(len(alignments), world, rank)

    if trim_fn is not None:
        local = []
        for i in mine:
            t = trim_fn(alignments[i])
            res, seq = getattr(t, "_res_mask", None), getattr(t, "_seq_mask", None)
            if res is None or seq is None:
                res, seq = t.residues_mask, t.sequences_mask
            local.append((np.asarray(res, dtype=bool), np.asarray(seq, dtype=bool), t))
    else:
        from . import _lib

        prepared = [trimmer._prepare(alignments[i], matrix) for i in mine]
        _mark("prepare")
        index = device.index if isinstance(device, torch.device) and device.index is not None else None
        if index is None:
            index = int(os.environ.get("PYTRIMAL_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        todo = [k for k, (_, dense, _, _, _) in enumerate(prepared) if dense.shape[0] and dense.shape[1]]
        results = {}
        packed_masks = None
        if todo:
            items = [(prepared[k][1], prepared[k][2], prepared[k][3]) for k in todo]
            for attempt in range(3):
                batch = _native_batch(index, max(1, min(int(threads), 64)))
                try:
                    outs = batch.trim(items)
                    break
                except _lib.BatchClosed:  # a thread asking for another worker count replaced the device's batch object
                    if attempt == 2:
                        raise
            for k, out in zip(todo, outs):
                if out[3] != _lib.OK:
                    batch.check(out[3], out[2])
                results[k] = out
            if len(todo) == len(prepared):  # (no empty alignment in between: the library's mask vector IS the gather's payload)
                packed_masks = getattr(outs, "packed", None)
        local = []
        for k, (names, dense, indet, params, _keep) in enumerate(prepared):
            if k in results:
                res, seq, info, _, rows = results[k]
            else:  # an empty alignment never reaches the device
                res, seq, info, rows = np.ones(dense.shape[1], dtype=bool), np.ones(dense.shape[0], dtype=bool), None, None
            if masks_only:
                if info is not None and info.warnings:
                    _raise_warnings(info, names, rows)
                t = None
            else:
                t = trimmer._finish(names, dense, alignments[mine[k]]._datatype, res, seq, info, rows, None, params)
            local.append((res, seq, t))
    _mark("native batch")
    collect = distributed and (world > 1 or force_collectives)
    if masks_only and not collect:
        return [(np.asarray(r, dtype=bool), np.asarray(s, dtype=bool)) for r, s, _ in local]
    if not collect:
        # (what the workers produced, as it is: rebuilding 64 results from their masks in the calling thread was a serial
        # tail of ~4 ms behind a 35 ms batch)
        return [t if isinstance(t, TrimmedAlignment) else _rebuild(alignments[i], r, s, _gap_stats(trimmer)) for i, (r, s, t) in zip(mine, local)]
    mine_trimmed = {i: t for i, (_, _, t) in zip(mine, local)}

    # every rank knows every shape, so shard payload sizes are known without a size exchange
    sizes = [(len(a.residues), len(a.sequences)) for a in alignments]
    width = max(max((sum(n + m for n, m in sizes[r::world]) for r in range(world)), default=0), 1)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    box = _gather_buffers(device, world, rank, width)
    # the shard's masks packed straight into the (page-locked) staging vector, one copy to the device, ONE gather into one
    # [world][width] tensor, one copy back for all ranks (round 5: a fresh device tensor, a pageable copy, a tensor and a
    # `.cpu()` per rank -- 64 x 5 KB, so every one of these is latency: profiles/r06_c5_collective.jsonl)
    stage = box.stage_np
    if trim_fn is None and packed_masks is not None and packed_masks.size <= stage.size:
        stage[:packed_masks.size] = packed_masks
    else:
        pos = 0
        for r, s, _ in local:
            stage[pos:pos + len(r)] = r
            pos += len(r)
            stage[pos:pos + len(s)] = s
            pos += len(s)
    if box.send is not box.stage:
        box.send.copy_(box.stage, non_blocking=True)
    _mark("pack + H2D")
    dist.gather(box.send, box.recv_list, dst=0, group=group)
    _mark("gather")
    if rank != 0:
        return None
    if box.recv_host is not box.recv:
        box.recv_host.copy_(box.recv, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
    flat_all = box.recv_host.numpy().view(np.bool_).copy()  # (the buffers are reused by the next call: one copy for all ranks)
    out = [None] * len(alignments)
    for r in range(world):
        flat = flat_all[r]
        pos = 0
        for i in range(r, len(alignments), world):
            n, m = sizes[i]
            res = flat[pos:pos + n]
            seq = flat[pos + n:pos + n + m]
            pos += n + m
            if masks_only:
                out[i] = (res, seq)
                continue
            t = mine_trimmed.get(i) if r == rank else None
            out[i] = t if isinstance(t, TrimmedAlignment) else _rebuild(alignments[i], res, seq, _gap_stats(trimmer))
    _mark("D2H + unpack")
    return out


# ---- the gather's buffers, kept between calls: one per (device, group size, rank, width) ----------------------------------
class _GatherBuffers:
    def __init__(self, device, world, rank, width):
        cuda = device.type == "cuda"
        self.stage = torch.zeros(width, dtype=torch.uint8, pin_memory=cuda)   # host side of the send buffer
        self.stage_np = self.stage.numpy()
        self.send = torch.zeros(width, dtype=torch.uint8, device=device) if cuda else self.stage
        self.recv = self.recv_host = self.recv_list = None
        if rank == 0:
            self.recv = torch.zeros((world, width), dtype=torch.uint8, device=device)
            self.recv_list = list(self.recv.unbind(0))
            self.recv_host = torch.zeros((world, width), dtype=torch.uint8, pin_memory=True) if cuda else self.recv


_GATHER = {}


def _gather_buffers(device, world, rank, width):
    # (the gather needs equally long tensors on every rank: every rank sizes its buffers by the same rule -- exactly `width`)
    key = (str(device), world, rank, width)
    box = _GATHER.get(key)
    if box is None:
        if len(_GATHER) >= 8:
            _GATHER.clear()
        box = _GATHER[key] = _GatherBuffers(device, world, rank, width)
    return box


# phase marks of one trim_batch call (tools/c5_collective.py): None = off
_TRACE = None


def _mark(what):
    if _TRACE is not None:
        import time
        _TRACE.append((what, time.perf_counter()))


def _gap_stats(trimmer):
    """Does this trimmer compute gap statistics (what its results' `terminal_only` shares with the source alignment)?"""
    from .trimmer import AutomaticTrimmer, ManualTrimmer

    return isinstance(trimmer, ManualTrimmer) or (isinstance(trimmer, AutomaticTrimmer) and trimmer.method != "noduplicateseqs")


def _rebuild(alignment, keep_res, keep_seq, gap_stats=False):
    dense = alignment._dense()
    whole = len(alignment._seq_idx) == len(alignment._names)  # (every sequence visible: the list of names is shared, not copied)
    out = TrimmedAlignment._from_parts(alignment._names if whole else alignment.names, dense, alignment._datatype, keep_seq, keep_res)
    out._gap_stats = gap_stats  # (what `terminal_only` counts over: trimmer._finish)
    return out
