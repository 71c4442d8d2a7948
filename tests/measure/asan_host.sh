#!/bin/bash
# Host-only half of the library (ingest, windows, cleaners: msastat_host.cpp) built with
# AddressSanitizer + UBSan by g++ and driven with corrupted FASTA/Clustal text. CPU only:
# sanitizers are not available for the GPU build on this pool.
set -e
cd "$(dirname "$0")/../.."
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC \
    -Iinclude pytrimal_amd/csrc/msastat_host.cpp -o /tmp/libmsahost_asan.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    ASAN_OPTIONS=detect_leaks=0 python3 tests/measure/asan_host.py
