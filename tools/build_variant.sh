#!/bin/bash
# Build a variant of the library with extra compiler flags into tools/abl/lib_<name>.so (git-ignored; the .so still
# travels to the GPU box with gpurun):   tools/build_variant.sh <name> [-DPMG_P2_SHAPE=4,4,16,2688 ...]
# Build options of the kernels: PMG_P<d>_SHAPE=bx,by,bz,max_m (patches.hpp), PMG_NWMAX_P/PMG_NWMAX_V,
# PMG_ITEM_P/PMG_ITEM_CW/PMG_ITEM_WPC, PMG_UNPAIRED_MASK, PMG_RING_P<d>, PMG_ABL_* (laplacian.hip).
cd "$(dirname "$0")/.."
mkdir -p tools/abl
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -fno-gpu-rdc "$@" \
  -o tools/abl/lib_$name.so pmg-dolfinx_amd/csrc/*.hip && echo built $name
