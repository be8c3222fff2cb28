#!/bin/bash
# Timing-only ablation builds of the large-K sweep (tools/bench_layer.py picks one with P4V_LIB=...): never shipped.
cd "$(dirname "$0")/.."
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form \
      -fno-slp-vectorize -DP4V_SW7_DBG=$d ptq4vit_amd/csrc/p4v_api.hip -o ptq4vit_amd/csrc/dbg/libp4v_sw7dbg$d.so &
done
wait
