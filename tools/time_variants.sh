#!/bin/bash
# Time the operator apply of every variant library in tools/abl (tools/build_variant.sh) for the P:n pairs given,
# on the GPU box:   tools/time_variants.sh 4:64 6:43 8:32
cd "$(dirname "$0")/.."
for lib in tools/abl/lib_*.so; do
  for Pn in "$@"; do
    PMG_AMD_LIB_ALLOW_MISSING=1 PMG_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/time_apply.py ${Pn%%:*} ${Pn##*:} 2>&1 | grep kernel \
      | sed "s|lib=[^ ]*|lib=$(basename $lib .so)|"
  done
done
