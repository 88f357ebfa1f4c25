#!/bin/bash
# Runs tools/rccl_capture_probe in its modes, each as its own process under a timeout; a mode that crashes is run
# once more under rocgdb for the backtrace.  Output: gpurun_out/rccl_capture_probe.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/rccl_capture_probe.txt
BIN=$ROOT/tools/rccl_capture_probe
export NCCL_DEBUG=${NCCL_DEBUG:-WARN}
# PROBE_LIBDIR: run against another copy of librccl / the HIP runtime (e.g. the one PyTorch bundles)
if [ -n "$PROBE_LIBDIR" ]; then export LD_LIBRARY_PATH=$PROBE_LIBDIR:$LD_LIBRARY_PATH; OUT=${OUT%.txt}_$(basename $(dirname $PROBE_LIBDIR)).txt; fi
: > $OUT; ldd $BIN | grep -i "rccl\|amdhip" >> $OUT
for mode in origin fork-nonblocking fork-blocking fork-eager-first; do
  timeout -k 5 60 $BIN $mode >> $OUT 2>&1
  rc=$?
  echo "== mode $mode: exit code $rc" >> $OUT
  if [ $rc -ne 0 ] && [ $rc -ne 2 ] && [ -z "$bt_done" ]; then
    echo "== backtrace (rocgdb) of mode $mode" >> $OUT
    timeout -k 5 120 /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args $BIN $mode 2>&1 | grep -v "^\[New Thread\|^\[Thread" | tail -60 >> $OUT
    bt_done=1
  fi
done
cat $OUT
