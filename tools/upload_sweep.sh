#!/bin/bash
# upload (pack + H2D) by number of packing helper threads and piece size; and the link's own rate from pinned memory
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for t in 0 1 3 5 7; do for kb in 512 1024 2048; do echo "pack helpers $t piece ${kb} KB"; MSA_PACK_THREADS=$t MSA_UPLOAD_PIECE_KB=$kb python tools/upload_time.py 2>/dev/null | head -2; done; done
python - <<'PY'
import torch, time
for mb in (4, 20, 25):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"pinned -> device {mb} MB: {dt*1e3:.3f} ms  {mb/1024/dt:.1f} GB/s")
PY
