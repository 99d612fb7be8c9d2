#!/bin/bash
# A/B build of ONE translation unit: tools/build_variant.sh NAME FILE.hip -DFLAG=... -> tabgnn_amd/libtabgnn_hip_NAME.so
# (the other objects come from the regular build; select it with TABGNN_LIB_PATH=<path>)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/models-for-relational-multimodal-data_amd"
NAME="$1"; FILE="$2"; shift 2
make -C "$PKG" -j8 >/dev/null
mkdir -p "$PKG/build/var_$NAME"
BASE="$(basename "$FILE" .hip)"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-strict-aliasing --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable "$@" \
  -c "$PKG/csrc/$BASE.hip" -o "$PKG/build/var_$NAME/$BASE.o"
OBJS=$(ls "$PKG"/build/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS "$PKG/build/var_$NAME/$BASE.o" -o "$PKG/tabgnn_amd/libtabgnn_hip_$NAME.so"
echo "$PKG/tabgnn_amd/libtabgnn_hip_$NAME.so"
