"""Turn gpurun_out/profile_<tag>/ (tools/profile_bench.sh) into the committed
summaries profiles/kernel_stats_<tag>.csv, profiles/hbm_traffic_<tag>.json and
profiles/README_<tag>.md."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"profile_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, f"kernel_stats_{tag}.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def counter(kind, name):
    f = glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            kn = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            agg[kn.split("(")[0]].append(float(r["Counter_Value"]))
    return agg


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
P, N, U = 4, 125, 64
cells_per_launch = 64**3 / 8
g_bytes = 48 * N * cells_per_launch  # the G stream of one launch: known byte count, read once, 16 B/lane
out = {"note": "KB counters of rocprofv3 --pmc, per launch (mean over launches). FETCH_SIZE counts a wide "
               "(16 B/lane) coalesced stream at half its bytes on gfx950 (MI355X_MICROARCH.md, HBM): the G stream "
               "(known byte count) is therefore added back once; the remaining reads (4/8-byte gathers, "
               "uncalibrated widths) are taken at face value."}
lines = ["| kernel | launches | FETCH_SIZE KB/launch | WRITE_SIZE KB/launch |", "|---|---|---|---|"]
for k in sorted(write):
    if k not in fetch:
        lines.append(f"| `{k.strip()[-60:]}` | {len(write[k])} | n/a (kernel absent from the fetch pass) | "
                     f"{sum(write[k]) / len(write[k]):.0f} |")
for k in sorted(fetch):
    fk = sum(fetch[k]) / len(fetch[k])
    if k not in write:  # the two passes must come from the same build (a renamed kernel has no partner): say so
        lines.append(f"| `{k.strip()[-60:]}` | {len(fetch[k])} | {fk:.0f} | n/a (kernel absent from the write pass) |")
        continue
    wk = sum(write[k]) / len(write[k])
    lines.append(f"| `{k.strip()[-60:]}` | {len(fetch[k])} | {fk:.0f} | {wk:.0f} |")
    # vector kernels: 16 B per lane streams both ways -- FETCH_SIZE at half its bytes, WRITE_SIZE exact
    if k.strip().startswith("ew_kernel2<Cheb") and "<true" in k:
        out.setdefault("vector_kernels_bytes_per_launch", {})[k.strip()] = {
            "fetch_kb_raw": fk, "write_kb": wk, "hbm_bytes_corrected": (2 * fk + wk) * 1024}
    if "stiffness_column_kernel<4, false" in k:  # <P, stored geometry[, streaming policy]>
        raw = (fk + wk) * 1024
        corrected = raw + g_bytes / 2
        out["stiffness_p4_fetch_kb_raw"] = fk
        out["stiffness_p4_write_kb"] = wk
        out["stiffness_p4_bytes_per_launch"] = corrected
        out["stiffness_p4_algorithmic_bytes_per_launch"] = (48 * N + 4 * N + 8 + 17 * U) * cells_per_launch
json.dump(out, open(os.path.join(dst, f"hbm_traffic_{tag}.json"), "w"), indent=1)

tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(dst, f"README_{tag}.md"), "w") as f:
    f.write(f"# rocprofv3 summary, round tag {tag}\n\n")
    f.write("Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu` "
            "(set-up kernels included), MI355X, ROCm 7.2.\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:16]:
        f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                f"{float(r['AverageNs'])/1e3:.2f} | {r['Percentage']} |\n")
    f.write("\n## HBM counters (separate --pmc passes)\n\n" + "\n".join(lines) + "\n\n")
    f.write("```json\n" + json.dumps(out, indent=1) + "\n```\n")
print(open(os.path.join(dst, f"README_{tag}.md")).read())
