#!/bin/bash
# Experimental build of the HIP library beside the product one: tools/build_variant.sh NAME "-DFLAG=..."  ->  hivemind_amd/csrc/variants/NAME.so
# (built in a scratch copy, so the in-tree objects and libhivemind_amd.so are not touched; select it with HIVEMIND_AMD_LIB=<path>).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/hm_variant_$NAME
mkdir -p $W/hivemind_amd/csrc $W/include $ROOT/hivemind_amd/csrc/variants
cp $ROOT/include/hivemind_amd.h $W/include/
cp $ROOT/hivemind_amd/csrc/*.hip $ROOT/hivemind_amd/csrc/*.hpp $ROOT/hivemind_amd/csrc/Makefile $W/hivemind_amd/csrc/
make -s -C $W/hivemind_amd/csrc -j6 EXTRA="$*" ${ONLY:+$ONLY}
cp $W/hivemind_amd/csrc/libhivemind_amd.so $ROOT/hivemind_amd/csrc/variants/$NAME.so
echo built $ROOT/hivemind_amd/csrc/variants/$NAME.so
