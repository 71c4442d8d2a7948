#!/bin/bash
# An A/B build of the library with extra -D switches, beside the shipped one:
#   bash tools/build_variant.sh NAME "-DMSA_ROW_U=8 ..."   ->  tools/_variants/NAME.so   (git-ignored; travels with gpurun's snapshot)
# tools/ab_variants.sh runs the measurements under every variant on the GPU box (it copies a variant over the library of the
# box's scratch snapshot; the tree here is never touched).
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/tools/_variants
make -s -C $ROOT/pytrimal_amd/csrc OBJDIR=build_$NAME OUT=$ROOT/tools/_variants/$NAME.so EXTRA="$*" -j4
rm -rf $ROOT/pytrimal_amd/csrc/build_$NAME
ls -la $ROOT/tools/_variants/$NAME.so
