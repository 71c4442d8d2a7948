#!/bin/bash
# SQ / TA / L2 counters of one kernel family under a PROGRAM (three rocprofv3 --pmc passes, kernel trace only).  What follows the
# kernel name goes straight behind `rocprofv3 ... --`, so its first word must be the interpreter or binary itself (python3, ./ubench):
# the profiler's preloaded library has initialised the GPU before the program starts, and `env X=1 ...`, `timeout ...`, `bash -c`,
# `taskset`, `numactl` or a `#!/usr/bin/env` script would be an exec hop behind that (refused on this pool).  Variables go in front
# of `bash tools/pmc_kernel.sh`.
#   bash tools/pmc_kernel.sh similarity_lg python3 tools/sim_once.py 1000 4000 2000
#   bash tools/pmc_kernel.sh similarity_lg python3 tools/c5_batch.py 4
# Prints per launch averages and the busy fractions of VALU, scalar unit, texture addresser / data, LDS, and the L2 hit rate.
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
K=$1; shift
case "$(basename "$1")" in env|timeout|bash|sh|taskset|numactl|nice|stdbuf) echo "pmc_kernel.sh: '$1' is a launcher, not the program: put variables in front of this script" >&2; exit 2;; esac
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rm -rf /tmp/pk1 /tmp/pk2 /tmp/pk3
( cd $R && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES --kernel-trace --output-format csv -d /tmp/pk1 -- "$@" > /dev/null 2>&1 )
( cd $R && rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pk2 -- "$@" > /dev/null 2>&1 )
( cd $R && rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d /tmp/pk3 -- "$@" > /dev/null 2>&1 )
python3 - "$K" <<'PY'
import csv, glob, collections, sys
key = sys.argv[1]
for d in ("/tmp/pk1", "/tmp/pk2", "/tmp/pk3"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        print(d, "no records"); continue
    v = {k: sum(x) / len(x) for k, x in acc.items()}
    n = len(next(iter(acc.values())))
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    print({k: "%.4g" % x for k, x in v.items()}, "launches", n)
    if "SQ_INSTS_VALU" in v:
        print("  kernel cycles %.3g  VALU busy %.2f  VALU/load %.2f  SALU/load %.2f  SALU per CU-cycle %.2f  LDS/load %.2f  waves %.0f" % (
            cyc, v["SQ_INSTS_VALU"] * 4 / (1024 * cyc), v["SQ_INSTS_VALU"] / v["SQ_INSTS_VMEM_RD"], v["SQ_INSTS_SALU"] / v["SQ_INSTS_VMEM_RD"],
            v["SQ_INSTS_SALU"] / (256 * cyc), v["SQ_INSTS_LDS"] / v["SQ_INSTS_VMEM_RD"], v["SQ_WAVES"]))
    if "TA_TA_BUSY_sum" in v:
        print("  TA busy %.2f  TD busy %.2f  LDS %.2f  mean waves per SIMD %.2f" % (v["TA_TA_BUSY_sum"] / 256 / cyc, v["TD_TD_BUSY_sum"] / 256 / cyc,
              v["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, v.get("SQ_WAVE_CYCLES", 0) / (1024 * cyc) if cyc else 0))
    if "TCC_HIT_sum" in v:
        print("  L2 hit rate %.3f  (hits %.3g misses %.3g)" % (v["TCC_HIT_sum"] / max(1, v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), v["TCC_HIT_sum"], v["TCC_MISS_sum"]))
PY
