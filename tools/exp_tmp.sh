cp pytrimal_amd/libmsastat_hip.so /tmp/shipped.so
export SIZES=209x1227,200x2000,300x3000,500x2000,500x5000,400x1000
for rep in 1 2; do
for v in shipped sort129; do
  if [ $v = shipped ]; then cp /tmp/shipped.so pytrimal_amd/libmsastat_hip.so; else cp tools/_variants/$v.so pytrimal_amd/libmsastat_hip.so; fi
  echo "== $v ($rep)"
  python tools/small_latency.py 2>&1 | grep -v amdgpu.ids | grep "strict" | cut -c1-200
done
done
cp /tmp/shipped.so pytrimal_amd/libmsastat_hip.so
