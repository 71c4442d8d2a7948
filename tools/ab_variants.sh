#!/bin/bash
# On the GPU box: the similarity kernel under every library variant of tools/_variants/ (tools/build_variant.sh), A B A B:
# cycle stamps and pass times at 1000 x 4000 and 2000 x 10000, the C5 batch, a cross-check against the sequential kernel.
#   bash tools/ab_variants.sh [out-dir] [variant names...]      (default: every variant, twice)
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$ROOT/gpurun_out/ab}; shift
mkdir -p $OUT
cd $ROOT
cp pytrimal_amd/libmsastat_hip.so /tmp/shipped.so
# (a timeout or an interrupt in the middle must not leave a -D variant installed as the library)
trap 'cp /tmp/shipped.so $ROOT/pytrimal_amd/libmsastat_hip.so' EXIT
NAMES="$@"; [ -z "$NAMES" ] && NAMES=$(ls tools/_variants/*.so | xargs -n1 basename | sed 's/\.so$//')
for rep in 1 2; do
for v in $NAMES; do
  if [ $v = shipped ]; then cp /tmp/shipped.so pytrimal_amd/libmsastat_hip.so; else cp tools/_variants/$v.so pytrimal_amd/libmsastat_hip.so; fi
  {
    echo "== $v (pass $rep)"
    timeout 120 python tools/bx_stamps.py 1000 4000 2000 2>/dev/null | grep sim_ms
    [ -z "$QUICK" ] && timeout 120 python tools/bx_stamps.py 2>/dev/null | grep sim_ms
    REPS=8 CHECK=$([ $rep = 1 ] && echo 1 || echo 0) timeout 300 python tools/sim_shapes.py 1000 4000 2000 2000 10000 1003 2>/dev/null
    [ -z "$QUICK" ] && timeout 300 python tools/c5_batch.py 4 2>/dev/null
    [ $rep = 1 ] && timeout 600 python tools/cross_check.py ${CASES:-12} 7 2>/dev/null | tail -3
  } >> $OUT/$v.txt 2>&1
done
done
cp /tmp/shipped.so pytrimal_amd/libmsastat_hip.so
for v in $NAMES; do echo "#### $v"; cat $OUT/$v.txt; done
