"""Where the waves of the similarity kernel run: per compute unit the columns it was dealt, their partner steps, and when its last
wave ended (stamped instantiation, MSA_SIM_MODE=64: record [5] = {XCC, SE / SH / CU}, [7] = the wave's end in 100 MHz ticks).
    python tools/cu_loads.py m n seed"""
import ctypes, json, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np, torch  # noqa: F401
from pytrimal_amd import _lib
from pytrimal_amd.synth import synth_msa
from bx_stamps import stamped_similarity
m, n, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
a = synth_msa(m, n, seed)
_, _, rec = stamped_similarity(a)
lib = _lib.load()
nw = min(rec["waves"], 16384)
buf = (ctypes.c_uint * (8 * nw))()
lib.msa_debug_bx_records.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.msa_debug_bx_records(buf, nw)
r = np.frombuffer(buf, dtype=np.uint32).reshape(nw, 8).astype(np.int64)
valid = (a != ord("-")) & (a != ord("X"))
# partner steps of a column: per 64-row round the valid rows at or behind its first row
suffix = np.cumsum(valid[::-1], axis=0)[::-1]
steps_col = suffix[0:m - 1:64].sum(0)
life = (r[:, 1] + r[:, 2] + r[:, 3]) * 64
end = r[:, 7] - r[:, 7].min()
start = end - life / (rec["clock_GHz"] * 10.0)
t0 = start.min()
end = end - t0
keys = r[:, 5]
cus = np.unique(keys)
rows = []
for k in cus:
    w = keys == k
    rows.append((int(k), int(w.sum()), int(steps_col[r[w, 0]].sum()), float(end[w].max()), float(life[w].mean())))
rows = np.array(rows)
out = {"m": m, "n": n, "waves": int(nw), "compute_units_seen": int(len(cus)), "clock_GHz": rec["clock_GHz"], "sim_ms": rec["sim_ms"],
       "waves_per_cu": {str(int(c)): int((rows[:, 1] == c).sum()) for c in np.unique(rows[:, 1])},
       "steps_per_cu": {"mean": float(rows[:, 2].mean()), "max": float(rows[:, 2].max()), "min": float(rows[:, 2].min())},
       "cu_end_ticks": {"mean": float(rows[:, 3].mean()), "max": float(rows[:, 3].max()), "min": float(rows[:, 3].min()),
                        "p10": float(np.percentile(rows[:, 3], 10)), "p90": float(np.percentile(rows[:, 3], 90))},
       "corr_steps_end": float(np.corrcoef(rows[:, 2], rows[:, 3])[0, 1]),
       "wave_end_ticks": {"mean": float(end.mean()), "max": float(end.max())},
       "wave_life_kcycles": {"mean": float(life.mean() / 1e3), "max": float(life.max() / 1e3)}}
print(json.dumps(out))
# the ten compute units that ended last and first
o = np.argsort(-rows[:, 3])
for i in list(o[:6]) + list(o[-4:]):
    print("cu %04x: %2d waves  %8d steps (%.2f of mean)  last wave ends at %6.0f ticks" % (int(rows[i, 0]), rows[i, 1], rows[i, 2], rows[i, 2] / rows[:, 2].mean(), rows[i, 3]))
