"""Per-position table of the child histogram launches from a rocprofv3 kernel trace (GPU box tool).

    python scripts/child_classes.py <trace_dir> [<pmc_dir> COUNTER]

Reads *kernel_trace.csv under <trace_dir>; a launch's class is its position between two root
launches (the node sizes repeat from tree to tree on the bench sets).  Prints, per position, the
launches, the average duration of k_hist_batch and -- in the same positions -- of the partition
and reduce + scan launches of the step, and the grid sizes.
"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = {}
    pos = {"k_hist_batch": -1, "k_decide_part": -1, "k_redscan": -1, "k_partition_batch": -1}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        if k == "k_hist_root":
            for n in pos:
                pos[n] = 0
            per.setdefault(("k_hist_root", 0), []).append(us)
            continue
        if k in pos and pos[k] >= 0:
            per.setdefault((k, pos[k]), []).append(us)
            pos[k] += 1
    names = sorted({k for k, _ in per})
    print("position " + " ".join(f"{n:>20s}" for n in names))
    for p in range(0, 12):
        if not any((n, p) in per for n in names):
            continue
        cells = []
        for n in names:
            v = per.get((n, p))
            cells.append(f"{sum(v) / len(v):12.1f} us x{len(v):3d}" if v else " " * 20)
        print(f"{p:8d} " + " ".join(cells))
    tot = {n: sum(sum(v) for (k, _), v in per.items() if k == n) for n in names}
    trees = len(per.get(("k_hist_root", 0), [1]))
    print("per tree " + " ".join(f"{tot[n] / trees:17.1f} us" for n in names))


if __name__ == "__main__":
    main()
