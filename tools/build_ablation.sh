#!/bin/bash
# Tuning / timing-only variant builds of the library into tools/abl/ (never used
# by tests or bench).  Usage: tools/build_ablation.sh NAME "-DFLAG ..." [NAME FLAGS ...]
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics $2 \
    -o tools/abl/libpmg_amd_$1.so pmg-dolfinx_amd/csrc/*.hip
  shift 2
done
