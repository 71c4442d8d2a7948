cp pytrimal_amd/libmsastat_hip.so /tmp/shipped.so
export SIZES=100x8000,200x6000,300x10000,500x8000,700x8000,1000x8000,1000x16000,64x20000
python tools/sim_shapes.py 500 8000 3 1000 8000 5 200 6000 7 2>&1 | grep -v amdgpu.ids | cut -c1-330
for rep in 1 2; do
for v in shipped n5120; do
  if [ $v = shipped ]; then cp /tmp/shipped.so pytrimal_amd/libmsastat_hip.so; else cp tools/_variants/$v.so pytrimal_amd/libmsastat_hip.so; fi
  echo "== $v ($rep)"
  python tools/small_latency.py 2>&1 | grep -v amdgpu.ids | grep "strict" | cut -c1-230
done
done
cp /tmp/shipped.so pytrimal_amd/libmsastat_hip.so
