cd $GRAFT_REPO_ROOT
cp pytrimal_amd/libmsastat_hip.so /tmp/shipped.so
for v in r04 shipped r04 shipped; do
  cp $([ $v = shipped ] && echo /tmp/shipped.so || echo tools/_variants/$v.so) pytrimal_amd/libmsastat_hip.so
  echo "#### $v"
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:r.get(k) for k in ('ms_per_step','ms_per_step_regions','ms_per_step_page_locked_rows','ms_per_step_resident','ms_per_step_public_api','kernels_ms')})"
done
cp /tmp/shipped.so pytrimal_amd/libmsastat_hip.so
