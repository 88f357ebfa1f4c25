#!/bin/bash
# One rocprofv3 --pmc pass per counter (group) of the set that did not return in round 1 (TA busy / TA
# stalled-by-TC / SQ LDS instruction counters), each under its own short timeout, on a SMALL workload
# (16^3 cells, p = 4, 2 repetitions), stopping at the first pass that fails or times out.
# Output: gpurun_out/pmc_probe/<name>.{log,rc}.  The program stands directly behind `--`.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { # name counters...
  local name=$1; shift
  echo "pass $name: $*" | tee -a $OUT/summary.txt
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/time_apply.py 4 16 2 > $OUT/$name.log 2>&1
  local rc=$?
  echo "  exit $rc" | tee -a $OUT/summary.txt
  return $rc
}
pass sq_lds SQ_ACTIVE_INST_LDS SQ_INSTS_LDS &&
pass ta_busy TA_TA_BUSY_sum &&
pass ta_addr TA_ADDR_STALLED_BY_TC_CYCLES_sum &&
pass ta_data TA_DATA_STALLED_BY_TC_CYCLES_sum
rc=$?
# Known to fail (profiles/pmc_probe_r02.md): three TA *_sum counters in one pass make rocprofv3 abort with error
# 38 and hang in its signal handler.  Only on explicit request, to re-check a new ROCm.
if [ $rc -eq 0 ] && [ "${PMC_PROBE_KNOWN_BAD:-0}" = "1" ]; then
  pass ta_all TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
  pass r1_group TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum SQ_ACTIVE_INST_LDS
fi
echo "done" | tee -a $OUT/summary.txt
