#!/bin/bash
# Tuning builds of the library into tools/abl/ (never used by tests or bench).
# Usage: tools/build_ablation.sh NAME "-DFLAG ..." [NAME FLAGS ...]
# Flags understood by laplacian.hip / patches.hpp (defaults in brackets):
#   -DPMG_NW=<max waves per workgroup>            [8; 4 at P = 3, 5, 6, 8]
#   -DPMG_WPS=<min waves per SIMD, P <= 4>  -DPMG_WPS_HI=<same, P >= 5>   [from NW; 1]
#   -DPMG_GDEPTH=<G layers in flight>             [1]
#   -DPMG_DLDS_FROM=<P from which the 1-D tables are re-read from LDS>     [9 = never]
#   -DPMG_COLUMN_MAX=<highest P on the column kernel; above: block kernel> [8]
#   -DPMG_BLOCK_WPS=<min waves per SIMD of the block kernel>               [1]
#   -DPMG_GFLAT_MASK=<bit P: flat, line-aligned G layout>                 [1<<2]
#   -DPMG_NO_NT                                   default cache policy instead of nt G loads / y stores
#   -DPMG_P4_BZ4, -DPMG_P1_SHAPE={bx,by,bz,cpr,max_m}, -DPMG_P2_SHAPE=..., -DPMG_P4_SHAPE=...  patch shapes
#   -DPMG_STAMPS                                  per-workgroup phase stamps (tools/stamp_phases.py)
# Select a build at run time with PMG_AMD_LIB=tools/abl/libpmg_amd_NAME.so.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics $2 \
    -o tools/abl/libpmg_amd_$1.so pmg-dolfinx_amd/csrc/*.hip
  shift 2
done
