#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MSA statistics path (BASELINE.json metric).

Workloads (BASELINE.json configs; `--workload`; default: C3 = the headline, at every N -- one alignment per GPU per step, weak
scaling, so that the lines of a 1 / 2 / 4 / 8 run hold the same per-GPU work; BASELINE config 5, the batch of 64 sharded over the
ranks, is timed behind it and rides in the same line as `c5_batch`):
  C3  AutomaticTrimmer('automated1') on a synthetic 2 000 x 10 000 protein MSA (seed 1003 + rank: all eight golden)
  C2  ManualTrimmer(gap_threshold=0.5, similarity_threshold=0.5) on 500 x 2 000 (seed 1002 + rank)
  C4  RepresentativeTrimmer(identity_threshold=0.5) on 5 000 x 5 000 (seed 1004 + rank)
  C5  the batch of 64 alignments of 1 000 x 4 000 (seeds 2000..2063), AutomaticTrimmer('automated1'), through
      `pytrimal_amd.batch.trim_batch`: sharded round-robin over the ranks (strong scaling), masks gathered over RCCL.
  REF the reference's own protocol (/root/reference/bench/bench.py:48-57,94-101): 3583 x 7287, the four statistic ->
      trimmer mappings, whole `trimmer.trim(alignment)` calls through the public API, median of 3.
One "step" = one pass of the whole path over the workload's input, FROM HOST ROWS: pack + H2D, kernels, D2H, host
selection logic -- SURVEY 8(d)'s metric, `value` / `ms_per_step`.  For C2-C4 the same steps with the residue bytes
already resident in HBM (`value_resident`) and through the public API (`AutomaticTrimmer(...).trim(Alignment)` ->
`TrimmedAlignment`: `value_public_api`) are timed right after.

`python bench.py --gpus N` starts N rank processes itself when it was not launched by torchrun (RANK unset): the
parent never touches a GPU, every child is `bench.py` again with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, a
child that fails makes the parent exit non-zero.  Ranks synchronise with a barrier + device synchronisation on both
sides of the timed steps, the time is the MAX over ranks, rank 0 prints ONE JSON line with `roofline` (dominant
kernel, HIP-event timed on the context's own stream) and, at N = 1, `cpu_baseline` (the CPU oracle, scalar and AVX2
flavours, on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_LANEOPS = 3.9e13  # 256 CUs x 64 lanes x 2.4 GHz (the ceiling of the pairwise passes)
# the similarity kernel's real ceiling: 256-byte coalesced dword-per-lane loads of L2-resident rows through a list of row
# offsets, 16 in flight, in EXACTLY the form the kernel issues (matrix base in an SGPR pair, list offset + lane offset by one
# v_add_u32 in a VGPR, hand-counted vmcnt), five waves per SIMD, measured with tools/ubench_wform.hip on an MI355X
# (profiles/r04_ubench_wform.txt, form 1): 29.4 TB/s chip-wide = 4.95 CU-cycles per wave-load at the 2.22 GHz the chip ran it at;
# every other address form (row base in SGPRs, 64-bit per-lane address by v_add_co / v_mad_u64_u32 / v_lshl_add_u64) measures
# the same within 2 %; with the kernel's three VALU instructions per step beside the loads 27.9 TB/s
W_STREAM_PEAK_GBS = 29400.0
W_STREAM_PEAK_WITH_VALU_GBS = 27900.0
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md: aggregate L2 -> L1 bandwidth figure of the guide


def similarity_w_stream_bytes(a, indet=ord("X")):
    """Bytes of W the similarity kernel moves through the vector-memory pipeline in one launch (one 256-byte row per
    partner step), computed from the alignment: per evaluated column and 64-row round, the valid rows at or behind
    the round's first row, in the round loop's blocks of 16 steps (the list is entered at a multiple of 16; nothing is
    requested behind a round's last block -- through round 4: blocks of 32, and 16 - 32 more loads of padding per round,
    which this count never included)."""
    m, n = a.shape
    valid = (a != ord("-")) & (a != indet)
    nvalid = valid.sum(axis=0)
    active = (a == ord("-")).sum(axis=0) / np.float32(m) < np.float32(0.8)
    steps = 0
    before = np.zeros(n, dtype=np.int64)
    for j0 in range(0, m - 1, 64):
        todo = nvalid - before
        live = active & (todo > 0)
        steps += int((((todo[live] + (before[live] & 15) + 15) // 16) * 16).sum())
        before = before + valid[j0:j0 + 64].sum(axis=0)
    return steps * 256, steps


def similarity_kernel_clock(a, vhash, dist):
    """The shader clock the similarity kernel runs at, GHz: one pass of the STAMPED kernel (MSA_SIM_MODE=64, a context of its own,
    behind every timed region) -- its waves' cycle counts (s_memtime) over their lifetimes in 100 MHz ticks (s_memrealtime).  A
    box whose chip clocks this kernel at 1.86 instead of 2.1 GHz explains most of a box-to-box difference of `value`."""
    from pytrimal_amd import _lib

    lib = _lib.load()
    lib.msa_debug_bx_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    saved = {k: os.environ.get(k) for k in ("MSA_SIM_MODE", "MSA_DIAGNOSTICS")}
    os.environ["MSA_SIM_MODE"] = "64"
    os.environ["MSA_DIAGNOSTICS"] = "1"  # (the switches are honoured only under this one)
    try:
        c = _lib.Context(0 if "LOCAL_RANK" not in os.environ else int(os.environ["LOCAL_RANK"]))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        buf = (ctypes.c_ulonglong * 16)()
        c.upload(a, ord("X"))
        c.similarity(vhash, dist)
        lib.msa_debug_bx_stamps(buf, 1)
        c.upload(a, ord("X"))
        c.similarity(vhash, dist)
        lib.msa_debug_bx_stamps(buf, 1)
    finally:
        c.close()
    return round((buf[0] + buf[1] + buf[2]) / max(buf[6], 1) / 10, 3) if buf[6] else None


WORKLOADS = {
    # name: (m, n, base seed)
    "C3": (2000, 10000, 1003),
    "C2": (500, 2000, 1002),
    "C4": (5000, 5000, 1004),
    "C5": (1000, 4000, 2000),
    "REF": (3583, 7287, 3583),
}
C5_BATCH = 64


def algorithmic_bytes(kernel, m, n):
    """every array touched once."""
    return {
        "gaps": m * n + 4 * n,
        "prep": 2 * m * n,
        "pairs": m * n + 8 * m * m,
        "sim": m * n + 4 * m * m + 8 * n,
        "encode": 2 * m * n,
        "idstats": 4 * m * m,
        "cluster": 4 * m * m,
        "overlap": m * n + 4 * m,
    }[kernel]


def launch_ranks(args):
    """--gpus N without a launcher: become the launcher.  Nothing here may touch a GPU (no HIP call, no
    torch.cuda.is_available(); counting devices does not initialise them on this image)."""
    if not args.launch_check and not args.share_gpu:
        import torch

        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            sys.exit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible; refusing to report a {args.gpus}-GPU number")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    codes = [p.wait() for p in procs]
    sys.exit(max(abs(c) for c in codes))


def launch_check(rank, world):
    """CPU-only check of the launcher (tests/test_distributed.py): gloo group, one all-reduce, one JSON line."""
    import torch
    import torch.distributed as dist

    dist.init_process_group(backend="gloo")
    t = torch.ones(1)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": int(t.item())}), flush=True)
    dist.destroy_process_group()


def host_info():
    info = {"nproc": os.cpu_count()}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            if line.startswith("Model name:"):
                info["cpu_model"] = line.split(":", 1)[1].strip()
            if line.startswith("Flags:"):
                flags = line.split(":", 1)[1].split()
                info["simd_flags"] = [f for f in ("sse2", "sse4_2", "avx", "avx2", "avx512f", "avx512bw") if f in flags]
    except Exception:
        pass
    try:
        info["git_sha"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                         timeout=10).stdout.strip() or None
    except Exception:
        info["git_sha"] = None
    if not info.get("git_sha"):  # (no .git on the GPU box: __graft_entry__.build() leaves the commit beside the library)
        try:
            with open(os.path.join(ROOT, "pytrimal_amd", "_build_info.json")) as f:
                info["git_sha"] = json.load(f).get("git_sha")
        except Exception:
            pass
    return info


def physical_cores():
    """Physical cores of the host (lscpu: sockets x cores per socket), or None."""
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        vals = {}
        for line in out.splitlines():
            k, _, v = line.partition(":")
            vals[k.strip()] = v.strip()
        return int(vals["Socket(s)"]) * int(vals["Core(s) per socket"])
    except Exception:
        return None


def cpu_baseline(a, trimmer_kw, ncols, value, with_similarity=True):
    """The CPU oracle on one host core (scalar port, then the AVX2 flavour of its two pairwise passes), and -- the
    reference's own idiom for many cores, README.md:136-152 -- a thread pool over WHOLE alignments, one per thread."""
    import oracle

    m, n = a.shape
    ncols = min(ncols, n)
    sample = np.ascontiguousarray(a[:, :ncols])
    pairs = m * (m - 1) // 2
    npass = 2 if with_similarity else 1  # pairwise passes of the workload (pair counts, similarity)
    t0 = time.perf_counter()
    oracle.trim(sample, **trimmer_kw)
    scalar_s = time.perf_counter() - t0
    flavours = {"scalar": {"kind": "port-scalar", "value": round(ncols / scalar_s, 2), "seconds": round(scalar_s, 2),
                           "pair_columns_per_s": round(npass * pairs * ncols / scalar_s, 1), "build": "gcc -O3 (auto-vectorised SSE2)"}}
    best = ("scalar", scalar_s)
    if oracle.lib_avx2() is not None:
        t0 = time.perf_counter()
        g, _, _, _ = oracle.gaps(sample)
        hit, dst = oracle.pair_counts(sample, avx2=True)
        w = oracle.weights(hit, dst)
        oracle.identities(hit, dst)
        if with_similarity:
            oracle.similarity(sample, w, g, *oracle.aa_matrix(), avx2=True)
        avx2_s = time.perf_counter() - t0
        flavours["avx2"] = {"kind": "port-avx2", "value": round(ncols / avx2_s, 2), "seconds": round(avx2_s, 2),
                            "pair_columns_per_s": round(npass * pairs * ncols / avx2_s, 1),
                            "build": "gcc -O3 -mavx2, intrinsics written from scratch (oracle/msa_oracle_avx2.c): gap counts + "
                                     "both pairwise passes (pair counts, similarity) + the two float matrices; the selection "
                                     "logic (< 1 % of a trim) is not included"}
        if avx2_s < best[1]:
            best = ("avx2", avx2_s)
    out = {
        "value": round(ncols / best[1], 2), "unit": "columns/s", "cores": 1, "kind": "port", "flavour": best[0],
        "sample": f"all {m} sequences x first {ncols} columns of the same alignment, one thread "
                  f"(cost per column equals the full workload's)",
        "seconds": round(best[1], 2),
        "pair_columns_per_s": round(npass * pairs * ncols / best[1], 1),
        "flavours": flavours, "host": host_info(),
        "note": "the reference's SIMD code cannot be built here (its trimAl submodule is empty); BASELINE.md quotes "
                "2.0e9 .. 6.3e9 pair-columns/s per laptop core for it",
    }
    speedups = {"speedup_vs_cpu_1core": round(value / (ncols / best[1]), 1)}
    cores = os.cpu_count() or 1
    if cores > 1:
        # the reference's batch idiom (README.md:136-152: ThreadPool.map(trimmer.trim, alignments), the trim releases the
        # interpreter lock): a pool over whole alignments, here one copy of the workload's first `per` columns per thread, as
        # many threads as the host has physical cores (at most 64)
        from multiprocessing.pool import ThreadPool

        phys = physical_cores()
        threads = max(2, min(phys or cores // 2, 64))
        per = min(1000, n)
        copies = [np.ascontiguousarray(a[:, :per]).copy() for _ in range(threads)]
        t0 = time.perf_counter()
        with ThreadPool(threads) as pool:
            pool.map(lambda x: oracle.trim(x, **trimmer_kw), copies)
        all_s = time.perf_counter() - t0
        out["all_cores"] = {"value": round(threads * per / all_s, 2), "unit": "columns/s", "cores": threads, "kind": "port-scalar",
                            "physical_cores": phys, "logical_cpus": cores,
                            "sample": f"a thread pool over {threads} whole alignments ({m} sequences x the first {per} columns each), one per "
                                      f"thread: the reference's ThreadPool.map(trimmer.trim, alignments) idiom; {all_s:.1f} s"}
        speedups["speedup_vs_cpu_all_cores"] = round(value / (threads * per / all_s), 1)
    return out, speedups


def run_reference_protocol(args, device):
    """The reference's own benchmark (bench/bench.py:48-57,94-101; README.md:156-160): 3583 sequences x 7287 columns,
    the four statistic -> trimmer mappings, `trimmer.trim(alignment)` timed with a wall clock around the whole call
    through the public API (host rows -> TrimmedAlignment), one untimed call first (the reference's first run of a fresh
    process is its cold one too), median of 3.  The reference's alignment (example.014.AA.EggNOG.COG0591.fasta) is not
    in its tree: the same shape from the seeded synthetic generator.  ONE JSON line: `value` = columns/s of the
    Similarity trim (the statistic BASELINE.md derives its 990 columns/s from), the four results under `results`."""
    from pytrimal_amd import Alignment, ManualTrimmer, OverlapTrimmer, RepresentativeTrimmer, _lib
    from pytrimal_amd.synth import synth_msa

    m, n, seed = WORKLOADS["REF"]
    a = synth_msa(m, n, seed)
    ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])
    # medians of 3 at 3583 sequences, AVX2 / SSE2 / platform=None, i7-10710U one core
    published = {"Gaps": (0.0052, 0.0055, 0.127), "Similarity": (7.34, 7.43, 75.9), "Overlap": (7.68, 9.84, 143.0),
                 "Identity": (8.57, 11.3, 131.9)}
    trimmers = {"Gaps": ManualTrimmer(gap_threshold=0.5, platform="hip"),
                "Similarity": ManualTrimmer(similarity_threshold=0.5, platform="hip"),
                "Overlap": OverlapTrimmer(60.0, 0.5, platform="hip"),
                "Identity": RepresentativeTrimmer(identity_threshold=0.5, platform="hip")}
    results = []
    for name, tr in trimmers.items():
        tr.trim(ali)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = tr.trim(ali)
            times.append(time.perf_counter() - t0)
        sec = float(np.median(times))
        ctx = _lib.thread_context()
        ctx.prof_enable(True)
        ctx.prof_reset()
        tr.trim(ali)
        kern = {}
        for nm in ("gaps", "prep", "pairs", "encode", "sim", "overlap", "cluster", "idstats"):
            ms, k = ctx.prof_get(nm)
            if k:
                kern[nm] = round(ms / k, 4)
        ctx.prof_enable(False)
        avx2, sse2, scalar = published[name]
        results.append({"statistic": name, "trimmer": repr(tr), "sequences": m, "residues": n, "times": [round(t, 6) for t in times],
                        "median": round(sec, 6), "columns_per_s": round(n / sec, 1), "kernels_ms": kern,
                        "kept": [int(sum(out.residues_mask)), int(sum(out.sequences_mask))],
                        "reference_published_median_s": {"avx2": avx2, "sse2": sse2, "platform_none": scalar,
                                                         "hardware": "i7-10710U @ 1.10 GHz, one core",
                                                         "data": "example.014.AA.EggNOG.COG0591.fasta (not in the reference tree)"},
                        "published_avx2_over_this": round(avx2 / sec, 1)})
    if args.out:
        with open(args.out, "w") as f:
            for r in results:
                f.write(json.dumps(r) + "\n")
    sim = next(r for r in results if r["statistic"] == "Similarity")
    print(json.dumps({
        "metric": "MSA columns/s (gap+similarity+identity)", "value": sim["columns_per_s"], "unit": "columns/s", "n_gpus": 1,
        "steps": 3, "warmup": 1, "ms_per_step": round(sim["median"] * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8/f32", "data": "synthetic",
        "config": {"workload": f"the reference's own protocol (bench/bench.py: four statistic -> trimmer mappings, whole trim() calls "
                               f"through the public API from host rows, median of 3) on a synthetic {m} seq x {n} col protein MSA; value = "
                               f"the Similarity trim (ManualTrimmer(similarity_threshold=0.5))", "m": m, "n": n},
        "results": results,
        "note": "BASELINE.md's published times are for the same shape on other data and other hardware (one laptop core): quoted "
                "beside each result, never as vs_baseline",
    }), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: C3 (the headline) at one GPU, C5 (BASELINE config 5, strong scaling) with --gpus N > 1")
    ap.add_argument("--out", default=None, help="REF: also write the four result lines to this file (JSON lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c5-leg", action="store_true",
                    help="C3 only: do not time BASELINE config 5 (the batch of 64 sharded over the ranks) behind the headline")
    ap.add_argument("--cpu-sample-cols", type=int, default=10000,
                    help="columns of the workload's alignment the CPU baseline is timed on (default: all of them at C3: ~7 s per flavour)")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing only: every rank on GPU 0, process group over gloo (RCCL needs a GPU per rank) -- runs the whole "
                         "multi-rank code path on a one-GPU box; the line it prints is marked and is not a multi-GPU measurement")
    args = ap.parse_args()
    if args.workload is None:
        # The headline (C3) at every N: one alignment per GPU per step -- `value` of an N-GPU line is then N x the same per-GPU work
        # as the 1-GPU line's (weak scaling: the two divide into an efficiency).  BASELINE config 5 -- the batch of 64 sharded over
        # the ranks, strong scaling -- is timed behind it at every N and reported in the same line as `c5_batch` (--no-c5-leg skips
        # it; `--workload C5` makes it the line's `value`).  Until round 6 a run with N > 1 defaulted to C5, so that the lines of a
        # 1 / 2 / 4 / 8 scaling run held two different workloads.
        args.workload = "C3"

    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args)  # does not return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        return launch_check(rank, world)

    import torch

    from pytrimal_amd import Alignment, AutomaticTrimmer, ManualTrimmer, RepresentativeTrimmer, _lib
    from pytrimal_amd.matrix import SimilarityMatrix
    from pytrimal_amd.synth import synth_msa

    if not torch.cuda.is_available() or _lib.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the product has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
        os.environ["PYTRIMAL_AMD_DEVICE"] = "0"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # under torchrun even a 1-rank job uses RCCL
        import torch.distributed as dist

        if args.share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    if args.workload == "REF":
        if world > 1:
            raise SystemExit("--workload REF is the reference's single-process protocol: run it with --gpus 1")
        return run_reference_protocol(args, device)

    m, n, seed = WORKLOADS[args.workload]
    matrix = SimilarityMatrix.aa()
    vhash = np.ascontiguousarray(matrix._vhash, dtype=np.int32)
    dmat = np.ascontiguousarray(matrix._dist, dtype=np.float32)

    def params_for(workload):
        P = _lib.TrimParams(0, -1.0, -1, -1.0, -1.0, -1, -1, -1, -1.0, -1.0, -1, -1.0, vhash.ctypes.data, dmat.ctypes.data,
                            len(matrix))
        if workload in ("C3", "C5"):
            P.method = _lib.METHOD_CODES["automated1"]
        elif workload == "C2":
            P.gap_threshold = float(np.float32(1) - np.float32(0.5))  # trimAlManager::gapThreshold = 1 - kwarg
            P.similarity_threshold = 0.5
        elif workload == "C4":
            P.max_identity = 0.5
        return P

    oracle_kw = {"C3": dict(method="automated1"), "C5": dict(method="automated1"),
                 "C2": dict(gap_threshold=0.5, similarity_threshold=0.5), "C4": dict(identity_threshold=0.5)}[args.workload]
    trimmer_repr = {"C3": "AutomaticTrimmer('automated1')", "C5": "AutomaticTrimmer('automated1')",
                    "C2": "ManualTrimmer(gap_threshold=0.5, similarity_threshold=0.5)",
                    "C4": "RepresentativeTrimmer(identity_threshold=0.5)"}[args.workload]

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if args.share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ctx = _lib.Context(local_rank)
    params = params_for(args.workload)
    kernels, resident_s, public_api_s, info, units_per_step, single_gpu_s = {}, None, None, None, None, None
    locked_s, cold_s, ranks_seen, region_ms, sim_clock = None, None, world, None, None
    if dist is not None:  # the ranks that really take part in the timed process group
        t = torch.ones(1, dtype=torch.float64, device="cpu" if args.share_gpu else device)
        dist.all_reduce(t)
        ranks_seen = int(t.item())

    def run_c5_batch(steps, warmup, with_public_api=True):
        """BASELINE config 5: the batch of 64 alignments of 1000 x 4000 through `trim_batch`, sharded round-robin over the ranks
        (strong scaling: the batch is the same whatever the number of ranks), from host rows to the masks gathered on rank 0.
        Returns seconds for `steps` steps (max over ranks), the same through the public objects, rank 0's one-GPU reference
        (world > 1) and the last step's masks."""
        from pytrimal_amd.batch import trim_batch

        m5, n5, seed5 = WORKLOADS["C5"]
        alis = []
        for k in range(C5_BATCH):
            a5 = synth_msa(m5, n5, seed5 + k)
            alis.append(Alignment([b"s%d" % i for i in range(m5)], [bytes(r) for r in a5]))
        trimmer = AutomaticTrimmer("automated1", platform="hip")

        def step(masks_only=True):
            # from host rows to the masks gathered on rank 0 (what the gather moves: BASELINE's "RCCL broadcast/gather of
            # trim masks"); the TrimmedAlignment objects of the public batch call are timed beside it (value_public_api)
            # (under a launcher even ONE rank gathers its masks through the process group: RCCL on a one-GPU box)
            return trim_batch(trimmer, alis, device=None if args.share_gpu else device, threads=4, masks_only=masks_only,
                              force_collectives=dist is not None)

        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        fence()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        public_api_s = None
        if with_public_api:
            step(False)
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(False)
            fence()
            public_api_s = max_over_ranks(time.perf_counter() - t0)
        # the same batch on ONE GPU (rank 0 alone, the others wait): the strong-scaling reference, so that a line of an
        # N-GPU run carries the 1-GPU number of the very same workload beside its own
        single_gpu_s = None
        if world > 1:
            if rank == 0:
                trim_batch(trimmer, alis, device=None if args.share_gpu else device, threads=4, shard=False, masks_only=True)
                t0 = time.perf_counter()
                for _ in range(max(1, min(steps, 5))):
                    trim_batch(trimmer, alis, device=None if args.share_gpu else device, threads=4, shard=False, masks_only=True)
                single_gpu_s = (time.perf_counter() - t0) / max(1, min(steps, 5))
            fence()
        kept = int(sum(int(res.sum()) for res, _ in out)) if rank == 0 else None
        return elapsed, public_api_s, single_gpu_s, kept

    c5_leg = None
    if args.workload == "C5":
        units_per_step = C5_BATCH * n  # the whole batch, however many ranks share it: strong scaling
        elapsed, public_api_s, single_gpu_s, kept = run_c5_batch(args.steps, args.warmup)
        # kernel times of one alignment of the batch (profiled separately: the batch runs on per-thread contexts)
        a = synth_msa(m, n, seed + rank)
        ctx.prof_enable(True)
        for _ in range(3):
            ctx.upload(a, ord("X"))
            _, _, info = ctx.trim(params)
        ctx.prof_enable(False)
    else:
        a = synth_msa(m, n, seed + rank)
        ld = (n + 63) // 64 * 64
        dev = torch.zeros((m, ld), dtype=torch.uint8, device=device)
        dev[:, :n] = torch.from_numpy(a).to(device)
        torch.cuda.synchronize()
        cdev = "cpu" if args.share_gpu else device  # where the collectives' tensors live
        gathered = [torch.empty(n, dtype=torch.uint8, device=cdev) for _ in range(world)] if rank == 0 else None
        units_per_step = world * n  # one alignment per rank per step: weak scaling
        # the rank's kept-column mask on its way to rank 0: one page-locked staging vector and one device vector for the whole
        # run (a tensor per step from pageable memory: an allocation and a staged copy per step)
        mask_host = torch.empty(n, dtype=torch.uint8).pin_memory() if dist is not None and not args.share_gpu else None
        mask_dev = torch.empty(n, dtype=torch.uint8, device=device) if mask_host is not None else None

        def finish(keep_res):
            if dist is None:
                return
            if mask_host is None:  # (--share-gpu: gloo, host tensors)
                dist.gather(torch.from_numpy(keep_res.view(np.uint8)), gathered, dst=0)
                return
            mask_host.numpy()[:] = keep_res.view(np.uint8)
            mask_dev.copy_(mask_host, non_blocking=True)
            dist.gather(mask_dev, gathered, dst=0)

        def step():
            # attach drops every derived buffer: each step recomputes the whole path from the bytes
            ctx.attach(dev.data_ptr(), m, n, ld, ord("X"))
            keep_res, keep_seq, info = ctx.trim(params)
            finish(keep_res)
            return keep_res, keep_seq, info

        def step_host_rows(rows=a, pin=False):
            # host rows -> device: ordinary (pageable) rows, as a numpy array holds them -- nothing registered, nothing staged by
            # the caller; the upload is enqueued without a wait of its own: the trim behind it waits for the stream once (the
            # rows outlive the step).  (Until round 4 the headline page-locked the rows once, in front of the timed steps; at
            # this size that no longer buys anything -- round 4: 3.395 against 3.352 ms -- and is now the number BESIDE the
            # headline: value_page_locked_rows.)
            ctx.upload(rows, ord("X"), pin=pin, wait=False)
            keep_res, keep_seq, info = ctx.trim(params)
            finish(keep_res)
            return keep_res, keep_seq, info

        trimmer_obj = {"C3": lambda: AutomaticTrimmer("automated1", platform="hip"),
                       "C2": lambda: ManualTrimmer(gap_threshold=0.5, similarity_threshold=0.5, platform="hip"),
                       "C4": lambda: RepresentativeTrimmer(identity_threshold=0.5, platform="hip")}[args.workload]()
        ali = Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a])

        def step_public_api():
            t = trimmer_obj.trim(ali)  # host rows -> TrimmedAlignment, the call a user of the reference makes
            finish(t._res_mask)
            return t

        for _ in range(args.warmup):
            step_host_rows()
        # THE TIMED REGION (value / ms_per_step): every step starts from the host rows -- SURVEY 8(d).
        # HIP event pairs around the two pairwise passes -- the dominant kernels, what `roofline` is computed from -- over
        # the timed region (level 2); the small kernels are timed in a few untimed steps afterwards: seven more event
        # pairs per step cost ~10 % of a 0.33 ms step (tools/step_overheads.py)
        ctx.prof_reset()
        ctx.prof_enable(2)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            keep_res, keep_seq, info = step_host_rows()
        fence()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        ctx.prof_enable(False)
        # four more regions of the same K steps, same fences: the median of the five says how much of `value` is the box's mood
        # (a region is 0.07 s at --steps 20, the clock is still ramping in the first)
        region_ms = [elapsed / args.steps * 1e3]
        for _ in range(4):
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_host_rows()
            fence()
            region_ms.append(max_over_ranks(time.perf_counter() - t0) / args.steps * 1e3)
        for name in ("pairs", "sim"):
            ms, launches = ctx.prof_get(name)
            if launches:
                kernels[name] = {"ms_avg": ms / launches, "launches": launches}
        ctx.prof_reset()
        ctx.prof_enable(True)
        for _ in range(5):
            step()
        fence()
        ctx.prof_enable(False)
        # the same steps with the residue bytes resident in HBM (no pack, no H2D)
        step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        resident_s = max_over_ranks(time.perf_counter() - t0)
        # ... and through the public API: Alignment -> trimmer.trim -> TrimmedAlignment (its own per-thread context)
        step_public_api()
        api_masks = step_public_api()  # (the alignment's rows are page-locked on its second trim: once, outside the timed steps)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_public_api()
        fence()
        public_api_s = max_over_ranks(time.perf_counter() - t0)
        assert np.array_equal(api_masks._res_mask, keep_res) and np.array_equal(api_masks._seq_mask, keep_seq)
        kept = int(info.kept_residues)
        # ... from PAGE-LOCKED rows (a copy of the alignment registered once, in front of its timed steps: every upload is then one
        # pitched DMA copy from where the rows lie), and COLD: the first trim of a fresh Alignment through the public API (pageable rows, a context that last saw
        # another alignment), one fresh object per sample
        locked = a.copy()
        step_host_rows(locked, pin=True)  # (registered here: msa_host_register, once per array)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_host_rows(locked, pin=True)
        fence()
        locked_s = max_over_ranks(time.perf_counter() - t0)  # (the page-locked leg, beside the pageable headline)
        cold = []
        for _ in range(5):
            fresh = Alignment(ali.names, [bytes(r) for r in a])
            fence()
            t0 = time.perf_counter()
            trimmer_obj.trim(fresh)
            cold.append(time.perf_counter() - t0)
        cold_s = max_over_ranks(float(np.median(cold)))
        if args.workload == "C3" and not args.no_c5_leg:
            # BASELINE config 5 behind the headline, at every N: the batch of 64 sharded over the ranks (strong scaling)
            k5 = max(1, min(args.steps, 10))
            e5, _, single5, kept5 = run_c5_batch(k5, max(1, min(args.warmup, 2)), with_public_api=False)
            c5_leg = {"seconds": e5, "steps": k5, "single_gpu_s": single5, "kept": kept5}

    if args.workload in ("C2", "C3") and rank == 0:
        sim_clock = similarity_kernel_clock(a, vhash, dmat)
    for name in ("prep", "pairs", "idstats", "gaps", "encode", "sim", "overlap", "cluster"):
        ms, launches = ctx.prof_get(name)
        if launches and name not in kernels:  # (pairs / sim of a resident workload: already taken from the timed region)
            kernels[name] = {"ms_avg": ms / launches, "launches": launches}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = units_per_step * args.steps / elapsed  # columns/s over the whole job
        top = dict(kernels)
        dom = max(top, key=lambda k: top[k]["ms_avg"] * top[k]["launches"]) if top else None
        roofline = None
        pairs = m * (m - 1) // 2
        if dom:
            alg = algorithmic_bytes(dom, m, n)
            achieved = alg / (kernels[dom]["ms_avg"] * 1e-3) / 1e9
            traffic, traffic_source = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                traffic = tj.get(f"{args.workload}:{dom}")
                traffic_source = tj.get("_source", "profiles/traffic.json (builder PMC pass, not measured in this run)")
            # what really bounds the dominant kernel (`achieved` / `peak` / `frac` stay the HBM figures of the contract:
            # algorithmic bytes per pass over the pass's time against the HBM peak)
            bound = {"sim": "vector-L1 / texture addresser (256-byte wave-loads of L2-resident W rows: roofline.w_stream); HBM carries "
                            "0.2 % of its peak",
                     "pairs": "VALU issue (bit-sliced compare + popcount: roofline.valu)"}.get(dom, "hbm")
            roofline = {
                "kernel": dom, "bound": bound, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes": alg, "ms_avg": round(kernels[dom]["ms_avg"], 4),
            }
            if dom == "sim":
                # one similarity pass = this many launches of the kernel (six rounds each from 1800 rows on: the columns stay on
                # the same blocks of W); ms_avg and achieved are per PASS, a profiler's per-kernel average is per launch
                lib = _lib.load()
                lib.msa_debug_sim_launches.argtypes = [ctypes.c_void_p]
                roofline["kernel_launches_per_pass"] = max(1, int(lib.msa_debug_sim_launches(ctx.h)))
                # (what `rocprofv3 --stats` lists as the kernel's average duration -- while the launches of a pass run one after the
                # other, as they do at C2 / C3 / C5.  Round 6: at 2560 - 4608 rows with more columns than wave slots, and for tall
                # alignments with 512 columns or more, the columns of a pass run as TWO staggered halves on two streams -- an odd
                # number of launches whose durations overlap: their sum exceeds the pass, and `ms_avg` -- HIP events around the
                # whole sequence on the context's stream, the join included -- is the pass)
                one_sequence = max(1, -(-((m - 1 + 63) // 64) // 6)) if m >= 1800 else 1  # (a launch per six rounds from 1800 rows on)
                roofline["kernel_launches_overlap"] = bool(roofline["kernel_launches_per_pass"] > one_sequence)
                roofline["ms_avg_per_kernel_launch"] = round(kernels[dom]["ms_avg"] / roofline["kernel_launches_per_pass"], 4)
            if dom == "sim" and args.workload != "C5":
                # what bounds it: the stream of W rows through the L1 / texture-addresser pipeline (all L2 hits), not HBM
                wbytes, wsteps = similarity_w_stream_bytes(a)
                rate = wbytes / (kernels[dom]["ms_avg"] * 1e-3) / 1e9
                roofline["w_stream"] = {
                    "bound": "vector-memory pipeline (L1/TA), 256-byte global_load_dword wave-loads of L2-resident W rows "
                             "(texture addresser 0.86 busy, texture data 0.88, VALU 0.71, L2 hit rate 0.967 at 2000 x 10000: profiles/r06_pmc_sim.txt)",
                    "partner_steps": wsteps, "bytes": wbytes, "achieved": round(rate, 1), "peak": W_STREAM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(rate / W_STREAM_PEAK_GBS, 4),
                    "frac_of_peak_with_the_loop_valu_work": round(rate / W_STREAM_PEAK_WITH_VALU_GBS, 4),
                    "frac_of_l2_peak": round(rate / L2_PEAK_GBS, 4), "l2_peak": L2_PEAK_GBS,
                    "peak_source": "tools/ubench_wform.hip, form 1 = the address form the kernel ships (constant SGPR base + list offset added to the "
                                   "lane offset by v_add_u32, hand-issued), five waves per SIMD, on one MI355X (profiles/r04_ubench_wform.txt); "
                                   "not measured in this run",
                }
            if dom in ("sim", "pairs"):
                # the pairwise passes are VALU-issue work, not bandwidth: one "pair-column" = one (j, k, column) term
                pcs = pairs * n / (kernels[dom]["ms_avg"] * 1e-3)
                roofline["valu"] = {
                    "pair_columns_per_s": round(pcs, 1),
                    "lane_ops_peak_per_s": VALU_PEAK_LANEOPS,
                    "pair_columns_per_lane_op_peak": round(pcs / VALU_PEAK_LANEOPS, 4),
                    "note": ("order-preserving fp32 accumulation evaluated in parallel (binade-exact kernel with per-lane "
                             "grids, DESIGN.md section 5): three VALU instructions, one LDS row and one 256-byte W row per 64 "
                             "(row, partner) terms; bound by the W stream through the vector-memory pipeline (roofline.w_stream), "
                             "HBM is irrelevant (the path moves tens of MB)") if dom == "sim" else
                            "bit-sliced compare / popcount over 32 columns per word on the seven symbol planes + validity: 11 VALU "
                            "instructions per pair and word; VALU issue (0.82 - 0.88 busy at 5000 x 5000, profiles/r04_pmc_sq.txt), not bandwidth",
                }
        roofline_all = {}
        for kname, kv in kernels.items():
            ach = algorithmic_bytes(kname, m, n) / (kv["ms_avg"] * 1e-3) / 1e9
            roofline_all[kname] = {"ms_avg": round(kv["ms_avg"], 4), "achieved_GBs": round(ach, 2),
                                   "frac_of_hbm_peak": round(ach / HBM_PEAK_GBS, 5)}
        if args.workload == "C5":
            workload = (f"{trimmer_repr} on the batch of {C5_BATCH} synthetic {m} seq x {n} col protein MSAs (C5, seeds {seed}.."
                        f"{seed + C5_BATCH - 1}) through trim_batch(threads=4), sharded round-robin over {world} rank(s), from "
                        f"host rows to gathered masks")
        else:
            workload = (f"{trimmer_repr} on synthetic {m} seq x {n} col protein MSA ({args.workload}, seed {seed}+rank), one "
                        f"alignment per GPU per step, every step from host rows (pack + H2D, kernels, D2H, host selection) to masks")
        out = {
            "metric": "MSA columns/s (gap+similarity+identity)" + (" [--share-gpu: all ranks on ONE GPU over gloo, a code-path test]" if args.share_gpu else ""),
            "value": round(value, 2),
            "unit": "columns/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if args.workload == "C5" else "weak",
            "vs_baseline": None,
            "dtype": "u8/f32",
            "data": "synthetic",
            "config": {
                "workload": workload, "m": m, "n": n,
                "selected_method": {1: "gappyout", 2: "strict"}.get(info.selected_method) if info is not None else None,
                "avg_seq": round(float(info.avg_seq), 6) if info is not None else None,
                "max_seq": round(float(info.max_seq), 6) if info is not None else None,
                "kept_columns": kept, "ranks": world, "ranks_seen": ranks_seen,
                "backend": dist.get_backend() if dist is not None else None,  # "nccl" = RCCL; None: no process group (one rank, no launcher)
                "parallelism": (f"batch of {C5_BATCH} sharded over {world} rank(s) x 4 threads" if args.workload == "C5"
                                else f"replicas x{world} (alignment per rank)"),
            },
            "roofline": roofline,
            "roofline_all_kernels": roofline_all,
            "kernels_ms": {k: round(v["ms_avg"], 4) for k, v in kernels.items()},
            "kernels_ms_source": ("HIP events: pairs and sim over the timed (host-rows) region, the other kernels over 5 untimed steps after it"
                                  if args.workload != "C5" else "HIP events over 3 untimed trims of one alignment of the batch"),
        }
        if c5_leg is not None:
            m5, n5, seed5 = WORKLOADS["C5"]
            out["c5_batch"] = {
                "workload": (f"BASELINE config 5: AutomaticTrimmer('automated1') on the batch of {C5_BATCH} synthetic {m5} seq x {n5} col protein "
                             f"MSAs (seeds {seed5}..{seed5 + C5_BATCH - 1}) through trim_batch(threads=4), sharded round-robin over {world} "
                             f"rank(s), from host rows to the masks gathered on rank 0"),
                "value": round(C5_BATCH * n5 * c5_leg["steps"] / c5_leg["seconds"], 2), "unit": "columns/s", "scaling": "strong",
                "ms_per_step": round(c5_leg["seconds"] / c5_leg["steps"] * 1e3, 4), "steps": c5_leg["steps"], "n_gpus": world,
                "kept_columns": c5_leg["kept"], "kept_columns_ok": c5_leg["kept"] == 192501,  # (tests/golden/configs.npz: the 64 masks)
                "note": "timed behind the headline with the same fences and the max over ranks; value(N) / value(1) of this object over "
                        "the lines of a 1 / 2 / 4 / 8 run is config 5's strong-scaling speed-up",
            }
            if c5_leg["single_gpu_s"]:
                out["c5_batch"]["same_batch_on_rank0_alone_ms"] = round(c5_leg["single_gpu_s"] * 1e3, 4)
        if args.workload == "C5" and world > 1 and single_gpu_s:
            out["strong_scaling_reference_1gpu"] = {
                "value": round(units_per_step / single_gpu_s, 2), "unit": "columns/s", "ms_per_step": round(single_gpu_s * 1e3, 4),
                "note": "the same batch of 64 on rank 0's GPU alone, timed behind the timed region: value / this = speed-up over one GPU"}
        if resident_s is not None:
            out["value_resident"] = round(units_per_step * args.steps / resident_s, 2)
            out["ms_per_step_resident"] = round(resident_s / args.steps * 1e3, 4)
        if public_api_s is not None:
            out["value_public_api"] = round(units_per_step * args.steps / public_api_s, 2)
            out["ms_per_step_public_api"] = round(public_api_s / args.steps * 1e3, 4)
        if args.workload != "C5":
            out["rows_page_locked"] = False  # (value / ms_per_step: ordinary pageable rows, nothing registered)
            out["ms_per_step_median_of_5_regions"] = round(float(np.median(region_ms)), 4)
            out["ms_per_step_regions"] = [round(x, 4) for x in region_ms]
            if sim_clock is not None:
                out["sim_kernel_clock_GHz"] = sim_clock
        if locked_s is not None:
            out["value_page_locked_rows"] = round(units_per_step * args.steps / locked_s, 2)
            out["ms_per_step_page_locked_rows"] = round(locked_s / args.steps * 1e3, 4)
        if cold_s is not None:
            out["value_cold"] = round(units_per_step / cold_s, 2)
            out["ms_cold"] = round(cold_s * 1e3, 4)
            out["value_cold_note"] = ("median of five FIRST trims, each of a fresh Alignment object through the public API: pageable rows, no "
                                      "registration (the Python layer page-locks an alignment's rows on its second trim: hipHostRegister, "
                                      "0.1 - 0.6 ms for 4 - 20 MB, profiles/r03_upload.txt)")
        if args.workload == "C5":
            out["config"]["kept_columns_expected"] = 192501  # (tests/golden/configs.npz: the 64 masks of config 5)
            out["config"]["kept_columns_ok"] = kept == 192501
        if not args.no_cpu_baseline and world == 1:
            sample = synth_msa(m, n, seed) if args.workload == "C5" else a
            ncols = args.cpu_sample_cols if args.workload != "C4" else min(args.cpu_sample_cols, 600)
            base, speedups = cpu_baseline(sample, oracle_kw, ncols, value, with_similarity=args.workload != "C4")
            out["cpu_baseline"] = base
            out.update(speedups)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
