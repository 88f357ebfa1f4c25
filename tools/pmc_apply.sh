#!/bin/bash
# Hardware counters of the p4 stiffness kernel (tools/time_apply.py 4 64), one rocprofv3 --pmc pass per
# group (run on the GPU box via gpurun).  Output: gpurun_out/pmc_apply/<group>/.
# The TA counters go one per pass: three TA `_sum` counters in one pass exceed the TA block's counter
# registers, rocprofv3 aborts ("error code 38: Request exceeds the capabilities of the hardware to
# collect") and then hangs in its own signal handler -- profiles/pmc_probe_r02.md.  Every pass runs
# under a timeout for that reason.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_apply
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" \
           "MemUnitStalled VALUBusy SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOT/tools/time_apply.py ${PMC_P:-4} ${PMC_N:-64} 5 > $OUT/g$i.log 2>&1
  rc=$?
  echo "group $i ($grp) exit $rc"
  [ $rc -ne 0 ] && break  # no further GPU step after a pass that failed or timed out
done
