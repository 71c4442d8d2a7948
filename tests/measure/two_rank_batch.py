"""Two ranks of `trim_batch` sharing ONE GPU (tests/test_gpu_configs.py::test_two_ranks_share_one_gpu): process group over
gloo -- a single-GPU box cannot run RCCL between two ranks -- but every trim goes through the native batch path on the
device: the sharding, the per-rank batch objects and the gather of the masks as a multi-GPU run does them.
   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/measure/two_rank_batch.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402

os.environ["PYTRIMAL_AMD_DEVICE"] = "0"  # both ranks on device 0
from pytrimal_amd import Alignment, AutomaticTrimmer  # noqa: E402
from pytrimal_amd.batch import broadcast_trimmer, trim_batch  # noqa: E402
from pytrimal_amd.synth import synth_msa  # noqa: E402

dist.init_process_group(backend="gloo")
rank = dist.get_rank()
shapes = [(40, 300), (200, 900), (64, 64), (333, 1200), (9, 77), (150, 2000), (500, 700)]
alis = []
for k, (m, n) in enumerate(shapes):
    a = synth_msa(m, n, 500 + k)
    alis.append(Alignment([b"s%d" % i for i in range(m)], [bytes(r) for r in a]))
trimmer = broadcast_trimmer(AutomaticTrimmer("automated1", platform="hip") if rank == 0 else None)
out = trim_batch(trimmer, alis, threads=2)
if rank == 0:
    single = [trimmer.trim(x) for x in alis]
    same = all(t.residues_mask == s.residues_mask and t.sequences_mask == s.sequences_mask and list(t.sequences) == list(s.sequences)
               for t, s in zip(out, single))
    print(json.dumps({"ranks": dist.get_world_size(), "alignments": len(out), "equal_to_single_process": bool(same),
                      "kept_columns": [int(sum(t.residues_mask)) for t in out]}), flush=True)
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
