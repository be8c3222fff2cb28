#!/bin/bash
# Timing-only ablation builds of the large-K sweep (tools/bench_layer.py picks one with P4V_LIB=...): never shipped.
cd "$(dirname "$0")/.."
# usage: build_ablation_libs.sh [sw6] <mask> ...   (sw6: ablations of k_sweep6 instead of k_sweep7)
K=SW7; if [ "$1" = "sw6" ]; then K=SW6; shift; elif [ "$1" = "sos" ]; then K=SOS; shift; elif [ "$1" = "sw8" ]; then K=SW8; shift; fi
mkdir -p ptq4vit_amd/csrc/dbg
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form \
      -fno-slp-vectorize -DP4V_${K}_DBG=$d ptq4vit_amd/csrc/p4v_api.hip -o ptq4vit_amd/csrc/dbg/libp4v_${K,,}dbg$d.so &
done
wait
