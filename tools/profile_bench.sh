#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + HBM counters of bench.py.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile_${1:-r01}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1; rc=$?; echo "stats exit $rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1; rc=$?; echo "fetch exit $rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1; rc=$?; echo "write exit $rc"; [ $rc -ne 0 ] && exit $rc
