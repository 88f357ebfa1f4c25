"""Kernel trace (rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/self_partner_cycle.py ...) reduced to
busy time per kernel family and idle time, over the last `cycles` of `total` equal cycles of the run.
usage: python tools/trace_families.py OUT cycles launches_per_cycle_hint"""
import csv, glob, os, re, sys
from collections import defaultdict

(path,) = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[:1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def family(n):
    n = re.sub(r"^void\s+", "", n)
    m = re.search(r"(stiffness_\w+_kernel<\d)", n)
    if m:
        return m.group(1) + ">"
    m = re.search(r"ew_kernel2<\(anonymous namespace\)::(\w+)", n)
    if m:
        return m.group(1)
    m = re.match(r"(?:\(anonymous namespace\)::|\w+::)*(\w+)", n)
    return m.group(1) if m else n[:40]


# a cycle = the period of the kernel-name sequence at the end of the trace
names = [family(n) for _, _, n in rows]
period = None
for p in range(20, len(names) // (ncyc + 1)):
    if names[-p:] == names[-2 * p:-p] and names[-p:] == names[-3 * p:-2 * p]:
        period = p
        break
if period is None:
    sys.exit("no periodic tail found")
sel = rows[-ncyc * period:]
span = sel[-1][1] - sel[0][0]
busy, count = defaultdict(int), defaultdict(int)
covered, cur = 0, sel[0][0]
for s, e, n in sel:
    busy[family(n)] += e - s
    count[family(n)] += 1
    if e > cur:
        covered += e - max(s, cur)
        cur = e
print(f"{period} kernels per cycle; {ncyc} cycles: span {span / ncyc / 1e6:.3f} ms per cycle, some kernel running "
      f"{covered / ncyc / 1e6:.3f} ms, idle {(span - covered) / ncyc / 1e6:.3f} ms ({100 * (span - covered) / span:.1f} %)")
for f, b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"  {f:42s} {count[f] / ncyc:6.1f} launches/cycle {b / ncyc / 1e3:9.1f} us/cycle {100 * b / span:5.1f} %  "
          f"({b / count[f] / 1e3:.1f} us each)")
