#!/bin/bash
# Diagnostic builds of libococc_hip.so with phases of occ_mlp_bwd_kernel compiled out (OCOCC_BWD_PHASES bit mask: 1 F0 + F1,
# 2 F2, 4 B2, 8 B1, 16 B0; results are garbage, the timing says what each phase costs).  Run on the build host, then
#   OCOCC_LIB_PATH=tools/probe/libococc_bwd_<mask>.so python tools/probe/decoder_train_bench.py
set -e
cd "$(dirname "$0")/../../objectcentricocccompletion_amd/csrc"
for mask in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -DOCOCC_BWD_PHASES=$mask \
    -c mlp_layer.hip -o /tmp/mlp_layer_$mask.o
  hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v mlp_layer.o) /tmp/mlp_layer_$mask.o \
    -o ../../tools/probe/libococc_bwd_$mask.so
done
