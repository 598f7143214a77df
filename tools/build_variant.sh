#!/bin/bash
# Build a timing / A-B variant of the library from the CURRENT sources with extra compiler flags:
#   tools/build_variant.sh NAME "-DSELFC_SOMETHING ..."   ->  selfc_amd/lib_NAME.so   (select with SELFC_LIB=selfc_amd/lib_NAME.so;
# tools/ab_libs.sh alternates bench runs between such libraries on one box).  Never shipped, never loaded by the tests.
set -eu
NAME=$1; FLAGS=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=/tmp/selfc_variant_$NAME
mkdir -p $OBJ
cd $ROOT/selfc_amd/csrc
SRCS="transforms dense_conv fused_gh fused_f fused_f16 stp backward dgrad_chain prof"
for f in $SRCS; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c $f.hip -o $OBJ/$f.o ) &
  if (( $(jobs -r | wc -l) >= 5 )); then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(for f in $SRCS; do echo $OBJ/$f.o; done) -o $ROOT/selfc_amd/lib_$NAME.so
echo "built selfc_amd/lib_$NAME.so with: $FLAGS"
