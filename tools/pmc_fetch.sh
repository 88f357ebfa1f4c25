#!/bin/bash
# HBM read traffic (FETCH_SIZE, KB per launch, raw) of the degree-4 stiffness kernel for the libraries in tools/abl named
# in LIBS (default: base touch4): does touching the tensor ahead fetch it twice?  One rocprofv3 --pmc pass per library.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_fetch
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for l in ${LIBS:-base touch4}; do
  PMG_AMD_LIB_ALLOW_MISSING=1 PMG_AMD_LIB=$ROOT/tools/abl/lib_$l.so timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$l -- python3 $ROOT/tools/time_apply.py 4 64 5 > $OUT/$l.log 2>&1
  rc=$?
  [ $rc -ne 0 ] && { echo "$l: exit $rc"; tail -5 $OUT/$l.log; break; }
  f=$(find $OUT/$l -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$l" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "stiffness_column_kernel<4" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
v = [float(r["Counter_Value"]) for r in rows]
print(f"{sys.argv[2]}: {len(v)} launches, FETCH_SIZE mean {sum(v)/max(len(v),1):.0f} KB per launch (raw)")
PY
done
