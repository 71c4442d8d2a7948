"""What the collective path of a multi-GPU C5 run costs a rank, measured on ONE GPU (round 6).
Under the launcher (the process group must exist before this process touches the GPU):
   python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tools/c5_collective.py [reps]
Legs, ms per call of `trim_batch(..., masks_only=True)` on the C5 alignments (seeds 2000 ...), four workers:
   * 64 alignments, no collectives (what `bench.py --workload C5` times at N = 1 without a launcher);
   * 64 alignments, one-rank `nccl` group, `force_collectives=True` (what the driver's N = 1 line under torchrun times);
   * 8 alignments (a rank's shard of an 8-GPU run) without and with the gather;
each split into the phases `pytrimal_amd.batch._TRACE` marks: prepare (the interpreter's per-alignment work in front of the native
call), native batch, pack + H2D, gather (RCCL), D2H + unpack; and the two fences `bench.py` puts around a timed region
(`dist.barrier()` + `torch.cuda.synchronize()`), timed by themselves."""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np
import torch
import torch.distributed as dist
from pytrimal_amd import Alignment, AutomaticTrimmer
from pytrimal_amd import batch as B
# (profiles/r06_c5_collective.jsonl also holds this tool's legs on round 5's collective path: `git show e3198f1:pytrimal_amd/batch.py`
# loaded as `B` in place of the module above, same box)
from pytrimal_amd.synth import synth_msa

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
alis = []
for k in range(64):
    a = synth_msa(1000, 4000, 2000 + k)
    alis.append(Alignment([b"s%d" % i for i in range(a.shape[0])], [bytes(r) for r in a]))
tr = AutomaticTrimmer("automated1", platform="hip")
plain = B.trim_batch(tr, alis, threads=4, masks_only=True)  # (no group yet: the library first, RCCL beside it afterwards)
torch.cuda.set_device(local_rank)
device = torch.device("cuda", local_rank)
grouped = "RANK" in os.environ
if grouped:
    dist.init_process_group(backend="nccl", device_id=device)
    dist.barrier()
    torch.cuda.synchronize()


def leg(name, sub, collect):
    rows = []
    for rep in range(reps + 2):
        B._TRACE = []
        t0 = time.perf_counter()
        out = B.trim_batch(tr, sub, device=device, threads=4, masks_only=True, force_collectives=collect, shard=collect)
        t1 = time.perf_counter()
        marks, B._TRACE = B._TRACE, None
        if rep < 2:
            continue
        ph = {}
        for (_, a), (what, b) in zip(marks, marks[1:]):
            ph[what] = ph.get(what, 0.0) + (b - a) * 1e3
        ph["total"] = (t1 - t0) * 1e3
        rows.append(ph)
    same = all(np.array_equal(o[0], p[0]) and np.array_equal(o[1], p[1]) for o, p in zip(out, plain))
    keys = list(rows[0])
    med = {k: round(statistics.median(r.get(k, 0.0) for r in rows), 4) for k in keys}
    best = round(min(r["total"] for r in rows), 4)
    print(json.dumps({"leg": name, "alignments": len(sub), "collectives": bool(collect), "backend": dist.get_backend() if grouped else None,
                      "ms_median_by_phase": med, "ms_total_best": best, "masks_equal_plain": bool(same)}), flush=True)
    return med["total"]


res = {}
for rnd in range(2):  # A B A B
    res.setdefault("c5_plain", []).append(leg("C5, no collectives", alis, False))
    if grouped:
        res.setdefault("c5_coll", []).append(leg("C5, one-rank nccl group, gather forced", alis, True))
    res.setdefault("s8_plain", []).append(leg("shard of 8, no collectives", alis[:8], False))
    if grouped:
        res.setdefault("s8_coll", []).append(leg("shard of 8, one-rank nccl group, gather forced", alis[:8], True))
if grouped:
    ts = []
    for _ in range(50):
        t = time.perf_counter()
        dist.barrier()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    print(json.dumps({"leg": "dist.barrier() + torch.cuda.synchronize()", "ms_median": round(statistics.median(ts), 4), "ms_best": round(min(ts), 4)}), flush=True)
    c5 = min(res["c5_plain"]), min(res["c5_coll"])
    s8 = min(res["s8_plain"]), min(res["s8_coll"])
    print(json.dumps({"summary": True, "c5_ms": c5[0], "c5_with_gather_ms": c5[1], "collective_overhead_ms_at_64": round(c5[1] - c5[0], 4),
                      "shard8_ms": s8[0], "shard8_with_gather_ms": s8[1], "collective_overhead_ms_at_8": round(s8[1] - s8[0], 4),
                      "projected_8gpu_speedup_over_1gpu": round(c5[0] / s8[1], 3), "projected_8gpu_efficiency": round(c5[0] / s8[1] / 8, 3)}), flush=True)
    dist.destroy_process_group()
