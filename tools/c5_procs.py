"""Is a batch of small alignments bound by the GPU or by something inside one process?  N processes x T threads, each
process trims its own list of 1000 x 4000 alignments; aggregate columns/s.
  python tools/c5_procs.py <processes> <threads> [alignments per process = 32]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    from pytrimal_amd import Alignment, AutomaticTrimmer
    from pytrimal_amd.batch import trim_batch
    from pytrimal_amd.synth import synth_msa
    threads, count = int(sys.argv[2]), int(sys.argv[3])
    alis = []
    for k in range(count):
        a = synth_msa(1000, 4000, 2000 + k)
        alis.append(Alignment([b"s%d" % i for i in range(1000)], [bytes(r) for r in a]))
    tr = AutomaticTrimmer("automated1", platform="hip")
    trim_batch(tr, alis, threads=threads)  # (also caches every alignment's detected type)
    print("ready", flush=True)
    sys.stdin.readline()
    t = time.perf_counter()
    for _ in range(3):
        trim_batch(tr, alis, threads=threads)
    print("done %f" % ((time.perf_counter() - t) / 3), flush=True)
    sys.exit(0)
procs, threads = int(sys.argv[1]), int(sys.argv[2])
count = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ps = [subprocess.Popen([sys.executable, __file__, "--worker", str(threads), str(count)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
      for _ in range(procs)]
for p in ps:
    assert p.stdout.readline().strip() == "ready"
t = time.perf_counter()
for p in ps:
    p.stdin.write("go\n"); p.stdin.flush()
secs = [float(p.stdout.readline().split()[1]) for p in ps]
wall = max(secs)
print({"processes": procs, "threads": threads, "alignments": procs * count, "batch_s_per_process": [round(s, 4) for s in secs],
       "columns_per_s": round(procs * count * 4000 / wall)})
