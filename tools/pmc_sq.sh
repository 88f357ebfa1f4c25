#!/bin/bash
# SQ-side counters of the stiffness kernel (where do the waves spend their cycles?), one rocprofv3 --pmc
# pass per group under a timeout; usage: PMC_P=4 PMC_N=64 tools/pmc_sq.sh   -> gpurun_out/pmc_sq_p<P>/
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=${PMC_P:-4}; N=${PMC_N:-64}
OUT=$ROOT/gpurun_out/pmc_sq_p$P
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOT/tools/time_apply.py $P $N 5 > $OUT/g$i.log 2>&1
  rc=$?
  echo "group $i ($grp) exit $rc"
  [ $rc -ne 0 ] && break
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "stiffness" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
with open(sys.argv[1] + "/summary.txt", "w") as out:
    for k in sorted(agg):
        line = f"{k:32s} launches {agg[k][0]:5d}  mean per launch {agg[k][1] / agg[k][0]:.4g}"
        print(line)
        out.write(line + "\n")
PY
