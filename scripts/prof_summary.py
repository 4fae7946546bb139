"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel table + one-iteration timeline."""
import csv
import sys

d = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else "bench"
rows = list(csv.DictReader(open(f"{d}/{pre}_kernel_stats.csv")))
print("| kernel | calls | total us | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['Name'].split('(')[0]} | {r['Calls']} | {int(r['TotalDurationNs'])/1e3:.1f} | "
          f"{float(r['AverageNs'])/1e3:.2f} | {int(r['MinNs'])/1e3:.2f} | {int(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
tr = list(csv.DictReader(open(f"{d}/{pre}_kernel_trace.csv")))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
# one iteration = from one lambda kernel to the next (the first launch of a boosting iteration)
idx = [i for i, r in enumerate(tr) if "k_lambda" in r["Kernel_Name"] or r["Kernel_Name"].startswith("k_residual")]
if len(idx) >= 3:
    a, b = idx[-3], idx[-2]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr[a:b])
    wall = int(tr[b]["Start_Timestamp"]) - int(tr[a]["Start_Timestamp"])
    print(f"\none iteration: wall {wall/1e3:.1f} us, kernels busy {busy/1e3:.1f} us, {b-a} launches")
    agg = {}
    for r in tr[a:b]:
        k = r["Kernel_Name"].split("(")[0]
        agg.setdefault(k, [0, 0])
        agg[k][0] += 1
        agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:24s} x{n:3d} {t/1e3:8.1f} us")
hist = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr if r["Kernel_Name"].startswith("k_hist")]
print("\nk_hist_root + k_hist launch durations of the last tree (us):", [round(x, 1) for x in hist[-10:]])
