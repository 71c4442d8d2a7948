cp pytrimal_amd/libmsastat_hip.so /tmp/shipped.so
for v in shipped nosload shipped nosload; do
  if [ $v = shipped ]; then cp /tmp/shipped.so pytrimal_amd/libmsastat_hip.so; else cp tools/_variants/$v.so pytrimal_amd/libmsastat_hip.so; fi
  echo "== $v"
  python tools/bx_stamps.py 1000 1000 11 2>&1 | grep -v amdgpu.ids
  python tools/bx_stamps.py 1000 4000 11 2>&1 | grep -v amdgpu.ids
  CHECK=0 python tools/sim_shapes.py 1000 300 11 1000 1000 11 1000 4000 11 2000 10000 1003 2>&1 | grep -v amdgpu.ids | cut -c1-120
done
cp /tmp/shipped.so pytrimal_amd/libmsastat_hip.so
