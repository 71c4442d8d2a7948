#!/bin/bash
# the randomised checks of a round, recorded as profiles/<tag>_fuzz.txt
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/fuzz.txt
: > $OUT
run() { echo -n "$* -> " >> $OUT; timeout 2400 "$@" 2>/dev/null | tail -1 >> $OUT; }
run python tests/fuzz/fuzz_trim.py ${1:-200} 707
run python tests/fuzz/fuzz_trim.py ${1:-200} 808
run python tests/fuzz/fuzz_trim.py ${3:-600} 909 tall
run python tools/cross_check.py 600 13
run python tests/fuzz/fuzz_threads.py ${2:-150} 4
run python tests/fuzz/fuzz_batch.py ${2:-150} 505
cat $OUT
