"""Where one V-cycle spends its time: the kernel trace of `pmg_main --cycles C` (rocprofv3 --kernel-trace csv)
reduced to busy time per kernel family and idle time between kernels, for the cycles only.
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- <repo>/pmg-dolfinx_amd/bin/pmg_main --n 64 --cycles 10
                  python tools/cycle_timeline.py OUT [cycles [fine-level stiffness launches per cycle]]"""
import csv, glob, os, re, sys
from collections import defaultdict

(path,) = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[:1]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed section of the driver: the last 11 cycles (1 warm-up + 10); a cycle starts at the fine level's first smoother kernel
names = [n for _, _, n in rows]
def family(n):
    m = re.search(r"(stiffness_column_kernel<\d)|(prolong_patch_kernel)|(restrict_patch_kernel)|ew_kernel2<\(anonymous namespace\)::(\w+)|(\w+_kernel)", n)
    if not m:
        return n[:40]
    return next(g for g in m.groups() if g)
# take the last fraction of the trace that holds `ncyc` repetitions of the per-cycle launch pattern
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 10
fine = [i for i, n in enumerate(names) if "stiffness_column_kernel<4" in n]
per_cycle_fine = int(sys.argv[3]) if len(sys.argv) > 3 else 7 * 8  # stiffness launches of the fine level per cycle
start = fine[-ncyc * per_cycle_fine]
# back up to the first kernel of that cycle (kernels between the previous cycle's last fine launch and this one)
prev = fine[-ncyc * per_cycle_fine - 1]
sel = rows[prev + 1:]
# drop trailing kernels after the last cycle's final kernel: keep through the last fine-level vector kernel
span = sel[-1][1] - sel[0][0]
busy = defaultdict(int); count = defaultdict(int)
idle = 0; last_end = sel[0][0]
for s, e, n in sel:
    f = family(n)
    busy[f] += e - s; count[f] += 1
    if s > last_end:
        idle += s - last_end
    last_end = max(last_end, e)
tot = sum(busy.values())
print(f"{len(sel)} kernels in {ncyc} cycles: span {span/ncyc*1e-6:.3f} ms per cycle, busy {tot/ncyc*1e-6:.3f} ms, idle between kernels {idle/ncyc*1e-6:.3f} ms ({100*idle/span:.1f} %)")
for f, t in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"  {f:42s} {count[f]/ncyc:7.1f} launches/cycle {t/ncyc*1e-3:9.1f} us/cycle {100*t/span:5.1f} %")
