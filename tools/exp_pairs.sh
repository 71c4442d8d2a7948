cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab6
for v in shipped xcd xcdprio; do
  [ $v != shipped ] && cp tools/_variants/$v.so pytrimal_amd/libmsastat_hip.so
  echo "#### $v"
  REPS=8 CHECK=0 python tools/sim_shapes.py 500 2000 1002 1000 4000 2000 2000 10000 1003 3000 8000 5 2>/dev/null | cut -c1-130
  python tools/c5_batch.py 4 2>/dev/null | cut -c1-110
  python tools/c5_batch.py 4 2>/dev/null | cut -c1-110
done
echo "#### hw queues (xcdprio build)"
for q in 2 4 8 16; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python tools/c5_batch.py 4 6 8 2>/dev/null | cut -c1-110; done
