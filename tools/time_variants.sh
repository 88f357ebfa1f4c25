#!/bin/bash
# Time the operator apply of variant libraries in tools/abl (tools/build_variant.sh) for the P:n pairs given, on the GPU
# box:   tools/time_variants.sh 4:64 6:43 8:32        (all libraries)
#        LIBS="wb stream4" tools/time_variants.sh 4:64   (only these)
cd "$(dirname "$0")/.."
libs=${LIBS:-$(ls tools/abl/lib_*.so | sed 's|tools/abl/lib_||; s|\.so||')}
for Pn in "$@"; do
  for l in $libs; do
    lib=tools/abl/lib_$l.so
    PMG_AMD_LIB_ALLOW_MISSING=1 PMG_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/time_apply.py ${Pn%%:*} ${Pn##*:} 2>&1 | grep kernel \
      | sed "s|lib=[^ ]*|lib=$l|"
  done
done
