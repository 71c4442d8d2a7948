export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES --kernel-trace --output-format csv -d /tmp/p1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/p2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("/tmp/p1","/tmp/p2"):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "similarity_lg" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    v={k:sum(x)/len(x) for k,x in acc.items()}
    cyc=v.get("GRBM_GUI_ACTIVE",0)/8
    print({k:"%.3g"%x for k,x in v.items()})
    if "SQ_INSTS_VALU" in v: print("VALU busy %.2f  VALU/load %.2f SALU/load %.2f  SALU per CU-cycle %.2f"%(v["SQ_INSTS_VALU"]*4/(1024*cyc), v["SQ_INSTS_VALU"]/v["SQ_INSTS_VMEM_RD"], v["SQ_INSTS_SALU"]/v["SQ_INSTS_VMEM_RD"], v["SQ_INSTS_SALU"]/(256*cyc)))
    if "TA_TA_BUSY_sum" in v: print("TA busy %.2f TD %.2f LDS %.2f"%(v["TA_TA_BUSY_sum"]/256/cyc, v["TD_TD_BUSY_sum"]/256/cyc, v["SQ_LDS_IDX_ACTIVE"]/256/cyc))
PY
