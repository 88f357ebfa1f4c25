#!/bin/bash
# Hardware counters of the p4 stiffness kernel (tools/time_apply.py 4 64), one rocprofv3 --pmc pass per
# group (run on the GPU box via gpurun).  Output: gpurun_out/pmc_apply/<group>/.
# (A pass with TA_TA_BUSY / TA_*_STALLED_BY_TC / SQ_ACTIVE_INST_LDS never returned on this pool: left out.)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_apply
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" \
           "MemUnitStalled VALUBusy SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOT/tools/time_apply.py 4 64 5 > $OUT/g$i.log 2>&1
  echo "group $i exit $?"
done
