#!/bin/bash
# builds tools/ubench_lstrip (loops generated into /tmp/lstrip_loops.h)
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
set -e
cd "$(dirname "$0")/.."
: > /tmp/lstrip_loops.h
python3 tools/gen_lstrip_loop.py 4 64 pk LS_4_64_PK >> /tmp/lstrip_loops.h
python3 tools/gen_lstrip_loop.py 4 64 fma LS_4_64_FMA >> /tmp/lstrip_loops.h
python3 tools/gen_lstrip_loop.py 8 64 pk LS_8_64_PK >> /tmp/lstrip_loops.h
python3 tools/gen_lstrip_loop.py 8 64 fma LS_8_64_FMA >> /tmp/lstrip_loops.h
python3 tools/gen_lstrip_loop.py 4 32 fma LS_4_32_FMA >> /tmp/lstrip_loops.h
python3 tools/gen_lstrip_loop.py 4 64 fma LS_4_64_FMA_NOFILL nofill >> /tmp/lstrip_loops.h
python3 tools/gen_lstrip_loop.py 4 64 fma LS_4_64_FMA_NOBAR nobarrier >> /tmp/lstrip_loops.h
python3 tools/gen_lstrip_loop.py 4 64 pk LS_4_64_PKG grouped >> /tmp/lstrip_loops.h
python3 tools/gen_lstrip_loop.py 4 64 fma LS_4_64_FMAG grouped >> /tmp/lstrip_loops.h
python3 tools/gen_lstrip_loop.py 8 64 pk LS_8_64_PKG grouped >> /tmp/lstrip_loops.h
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-inline-asm -Wno-unused-value -o tools/ubench_lstrip tools/ubench_lstrip.hip
