#!/bin/bash
# Tuning builds of the library into tools/abl/ (never used by tests or bench).
# Usage: tools/build_ablation.sh NAME "-DFLAG ..." [NAME FLAGS ...]
# Flags understood by laplacian.hip: -DPMG_NW=<waves per workgroup>,
# -DPMG_WPS=<min waves per SIMD>, -DPMG_GDEPTH=<G layers in flight>, -DPMG_P4_BZ4,
# -DPMG_STAMPS (per-workgroup phase stamps, read with tools/stamp_phases.py).
# Select a build at run time with PMG_AMD_LIB=tools/abl/libpmg_amd_NAME.so.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics $2 \
    -o tools/abl/libpmg_amd_$1.so pmg-dolfinx_amd/csrc/*.hip
  shift 2
done
