"""Per-position counter table of the histogram launches (GPU box tool).

    python scripts/pmc_child_table.py <out.md> <pass_dir> [<pass_dir> ...]

Every <pass_dir> is one `rocprofv3 --kernel-trace --pmc ...` run of the same bench command (counters
that do not fit one pass go into several).  The k_hist_batch dispatches are put into classes by their
position behind the last k_hist_root (the node sizes repeat from tree to tree on the bench set), every
counter is averaged per class over the trees of the run, and one row per class is printed with the
launch's duration under the counters."""
import csv
import glob
import sys


def rows_of(d):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    table, dur, order = {}, {}, []
    for d in dirs:
        per = {}
        for r in rows_of(d):
            k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
            c = r["Counter_Name"]
            per.setdefault(k, {}).setdefault(c, 0.0)
            per[k][c] += float(r["Counter_Value"])
            per[k]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        pos, trees = -1, 0
        for (did, name), cs in sorted(per.items()):
            if name.startswith("k_hist_root"):
                pos, trees = 0, trees + 1
                key = ("root", 0)
            elif name.startswith("k_hist_batch") and pos >= 0:
                key = ("child", pos)
                pos += 1
            else:
                continue
            if trees <= 1:       # (the first tree of a run: warm-up)
                continue
            for c, v in cs.items():
                if c == "_us":
                    dur.setdefault(key, []).append(v)
                    continue
                if c not in order:
                    order.append(c)
                table.setdefault(key, {}).setdefault(c, []).append(v)
    keys = sorted(table, key=lambda k: (k[0] != "root", k[1]))
    with open(out, "w") as f:
        f.write("| launch | n | us | " + " | ".join(order) + " |\n|---|---|---|" + "---|" * len(order) + "\n")
        for k in keys:
            us = dur.get(k, [0.0])
            cells = []
            for c in order:
                v = table[k].get(c)
                cells.append(f"{sum(v) / len(v):.4g}" if v else "")
            f.write(f"| {k[0]} {k[1]} | {len(us) // max(1, len(dirs))} | {sum(us) / len(us):.1f} | " + " | ".join(cells) + " |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
