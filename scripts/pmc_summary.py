"""Summarise rocprofv3 --pmc output for k_hist: per-dispatch counter values."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter_collection.csv under", d); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
print(len(rows), "rows; columns:", list(rows[0].keys()))
names = sorted({r["Counter_Name"] for r in rows})
print("counters:", names)
for name in names:
    vals = {}
    for r in rows:
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0]
        vals.setdefault(k, []).append(float(r["Counter_Value"]))
    for k, v in sorted(vals.items(), key=lambda kv: -max(kv[1])):
        if k.startswith("k_hist") or k.startswith("k_reduce") or "k_lambda" in k:
            v2 = sorted(v, reverse=True)
            print(f"{name:12s} {k:16s} n={len(v):4d} max={v2[0]:.1f} top5={[round(x,1) for x in v2[:5]]} sum={sum(v):.1f}")
