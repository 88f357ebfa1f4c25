R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr32_w -- python3 $R/tools/self_partner_cycle.py 14 eager 32 windows > $R/gpurun_out/tr32_w.log 2>&1
python3 $R/tools/trace_families.py $R/gpurun_out/tr32_w 10
python3 - $R/gpurun_out/tr32_w <<'PY'
import csv, glob, os, sys, collections
(path,) = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[:1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if "window_exchange" in r["Kernel_Name"]:
        d[(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Workgroup_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = v[len(v)//2:]
    print("grid", k, "n", len(v), "median %.1f us" % sorted(v)[len(v)//2])
PY
rm -rf $R/gpurun_out/tr32_w
