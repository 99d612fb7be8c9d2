#!/bin/bash
# Build A/B variants of the fused encoder kernels (EF_ABL bit flags, see csrc/encoder_fused.hip) as separate libraries
# under build/abl_<N>/ : tools/encoder_ablate.sh 0 1 2 4 8 32 ; then TABGNN_LIB_PATH=<lib> python tools/encoder_probe.py
set -e
cd "$(dirname "$0")/../models-for-relational-multimodal-data_amd"
OBJ=$(ls build/*.o | grep -v encoder_fused)
for n in "$@"; do
  mkdir -p build/abl_$n
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DEF_ABL=$n ${EF_EXTRA} -c csrc/encoder_fused.hip -o build/abl_$n/encoder_fused.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJ build/abl_$n/encoder_fused.o -o build/abl_$n/libtabgnn_hip.so
done
