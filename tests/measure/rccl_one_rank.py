"""RCCL in the same process as libmsastat_hip.so, on ONE GPU (tests/test_gpu_configs.py::test_rccl_one_rank_collectives).
A one-rank NCCL (= RCCL on ROCm) process group is created AFTER the HIP library has been loaded and has computed on the
device (three HIP users in one process: torch's bundled runtime, its bundled librccl, libmsastat_hip.so -- the load order
pytrimal_amd/_lib.py documents), then the calls a multi-GPU run of `trim_batch` makes go over it: `broadcast_trimmer`
(`broadcast_object_list`), an `all_reduce` of a device tensor, and `trim_batch(..., force_collectives=True)`, whose
`dist.gather` of the packed uint8 device buffer executes even at world size 1.  The masks that come back through the
collective are compared with the ones the same trimmer computes alone, and with the CPU oracle for the smallest alignment.
   python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tests/measure/rccl_one_rank.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pytrimal_amd import Alignment, AutomaticTrimmer, _lib  # noqa: E402
from pytrimal_amd.batch import broadcast_trimmer, trim_batch  # noqa: E402
from pytrimal_amd.synth import synth_msa  # noqa: E402

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
shapes = [(40, 300), (200, 900), (333, 1200), (1000, 4000), (9, 77)]
alis, dense = [], []
for k, (m, n) in enumerate(shapes):
    a = synth_msa(m, n, 700 + k)
    dense.append(a)
    alis.append(Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a]))

# the HIP library first: loaded, a context created, a trim computed -- before RCCL exists in this process
alone = AutomaticTrimmer("automated1", platform="hip")
single = [alone.trim(x) for x in alis]
loaded_before = _lib.load() is not None

torch.cuda.set_device(local_rank)
device = torch.device("cuda", local_rank)
dist.init_process_group(backend="nccl", device_id=device)
rank, world = dist.get_rank(), dist.get_world_size()
t = torch.full((4,), float(rank + 1), dtype=torch.float64, device=device)
dist.all_reduce(t)
torch.cuda.synchronize()
trimmer = broadcast_trimmer(AutomaticTrimmer("automated1", platform="hip") if rank == 0 else None)
masks = trim_batch(trimmer, alis, device=device, threads=2, masks_only=True, force_collectives=True)
objects = trim_batch(trimmer, alis, device=device, threads=2, force_collectives=True)
dist.barrier()
torch.cuda.synchronize()
# ... and the HIP library again with RCCL alive beside it
again = [trimmer.trim(x) for x in alis]

import oracle  # noqa: E402  (the checker)

res, seq, _ = oracle.trim(dense[0], method="automated1")
same_masks = all(np.array_equal(r, np.asarray(s.residues_mask)) and np.array_equal(q, np.asarray(s.sequences_mask))
                 for (r, q), s in zip(masks, single))
same_objects = all(o.residues_mask == s.residues_mask and o.sequences_mask == s.sequences_mask and list(o.sequences) == list(s.sequences)
                   for o, s in zip(objects, single))
same_again = all(o.residues_mask == s.residues_mask and o.sequences_mask == s.sequences_mask for o, s in zip(again, single))
oracle_ok = bool(np.array_equal(masks[0][0], res.astype(bool)) and np.array_equal(masks[0][1], seq.astype(bool)))
print(json.dumps({"backend": dist.get_backend(), "world": world, "all_reduce": float(t[0].item()), "hip_library_loaded_first": bool(loaded_before),
                  "trimmer_repr": repr(trimmer), "trimmer_platform": trimmer.platform, "gathered_masks_equal_single": bool(same_masks), "gathered_objects_equal_single": bool(same_objects),
                  "trims_after_rccl_equal": bool(same_again), "oracle_equal": oracle_ok,
                  "kept_columns": [int(np.sum(r)) for r, _ in masks]}), flush=True)
dist.destroy_process_group()
