#!/bin/bash
# Timing-only ablation builds of the library (results are WRONG by construction;
# never used by tests or bench).  PMG_ABL=4: memory skeleton of the stiffness
# kernel (all loads/stores, no contraction).  Output: tools/abl/libpmg_amd_abl<k>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
for k in 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DPMG_ABL=$k \
    -o tools/abl/libpmg_amd_abl$k.so pmg-dolfinx_amd/csrc/*.hip
done
