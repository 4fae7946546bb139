"""Turn rocprofv3 --pmc passes into the JSON files bench.py and DESIGN.md quote (GPU box tool).

    python scripts/pmc_tables.py hist   <dir_fetch> <dir_write> <out.json>
        per-dispatch FETCH_SIZE / WRITE_SIZE of k_hist_root and k_hist_batch (two separate passes
        of the same command, as MI355X_MICROARCH.md prescribes).  The child launches are put into
        CLASSES by their position in the tree (their order between two root launches: the
        node sizes repeat from tree to tree on the bench set), so that the bytes can be attributed:
        every class with its launches, FETCH / WRITE KB per launch, and -- given the tree shapes in
        <dir_fetch>/shapes.json if present -- its documents.
    python scripts/pmc_tables.py lambda <dir_sq> <out.json>
        SQ_INSTS_VALU / SALU / LDS and SQ_WAVES of k_lambda per launch and per query.
"""
import csv
import glob
import json
import os
import sys


def rows_of(d):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}")
    return list(csv.DictReader(open(f[0])))


def per_dispatch(rows, counter, with_us=False):
    """[(dispatch id, kernel, value[, duration in us])] in dispatch order for one counter."""
    out, dur = {}, {}
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
        out[k] = out.get(k, 0.0) + float(r["Counter_Value"])
        dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    if with_us:
        return sorted((d, k, v, dur[(d, k)]) for (d, k), v in out.items())
    return sorted((d, k, v) for (d, k), v in out.items())


def hist(dir_fetch, dir_write, out, millions=1):
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of "
                     "python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 6 --warmup 2"
                     + (f" --queries {10000 * millions}" if millions != 1 else ""),
           "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads "
                         "(MI355X_MICROARCH.md, HBM section) -> the ROOT launch's read bytes = 2 * FETCH_SIZE * 1024; "
                         "the child launches GATHER 48-byte rows (64-B requests), where the counter is taken at "
                         "face value: FETCH_SIZE * 1024 -- which reproduces the 334 B per gathered document that "
                         "the row layout predicts (3 blocks x 1.75 64-B lines); WRITE_SIZE * 1024 uncalibrated"}
    per = {}
    for name, d in (("FETCH_SIZE", dir_fetch), ("WRITE_SIZE", dir_write)):
        seq = per_dispatch(rows_of(d), name, with_us=True)
        # classes: position of a k_hist_batch dispatch after the last k_hist_root
        pos, trees = -1, 0
        for _, k, v, us in seq:
            if k.startswith("k_hist_root"):
                pos, trees = 0, trees + 1
                per.setdefault(("root", 0), {}).setdefault(name, []).append(v)
                per[("root", 0)].setdefault("us", []).append(us)
            elif k.startswith("k_hist_batch") and pos >= 0:
                per.setdefault(("child", pos), {}).setdefault(name, []).append(v)
                per[("child", pos)].setdefault("us", []).append(us)
                pos += 1
    root = per.get(("root", 0), {})
    if root:
        f = sum(root["FETCH_SIZE"]) / len(root["FETCH_SIZE"])
        w = sum(root["WRITE_SIZE"]) / len(root["WRITE_SIZE"])
        n = 1000000 * millions
        res.update({"kernel": f"k_hist_root (root launch, {millions}M docs x 136 features)", "FETCH_SIZE_KB": round(f, 1),
                    "WRITE_SIZE_KB": round(w, 1), "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
                    "algorithmic_bytes_per_launch": n * 136 + 8 * n + 136 * 256 * 16})
    classes = []
    tf = tw = 0.0
    ntrees = max(1, len(root.get("FETCH_SIZE", [1])))
    for (kind, pos), v in sorted(per.items()):
        if kind != "child":
            continue
        f, w = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
        classes.append({"position_in_tree": pos, "launches": len(f),
                        "FETCH_KB_per_launch": round(sum(f) / max(1, len(f)), 1),
                        "WRITE_KB_per_launch": round(sum(w) / max(1, len(w)), 1),
                        "us_per_launch_under_counters": round(sum(v["us"]) / max(1, len(v["us"])), 1)})
        tf += sum(f)
        tw += sum(w)
    res["child_launches"] = {"kernel": "k_hist_batch", "classes": classes,
                             "FETCH_MB_per_tree": round(tf / ntrees / 1024, 1),
                             "WRITE_MB_per_tree": round(tw / ntrees / 1024, 1), "trees": ntrees}
    try:   # (bench.py quotes these counters only for the kernel they were collected on)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from bench import hist_source_fingerprint
        res["kernel_source_fingerprint"] = hist_source_fingerprint()
    except Exception:
        pass
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def lam(d, out):
    rows = rows_of(d)
    res = {"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES of "
                     "python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 20 --warmup 5 "
                     "(10,000 queries of 100 documents; every query still holds tied scores in these iterations)"}
    for name, key in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_LDS", "lds"), ("SQ_WAVES", "waves")):
        v = [x for _, k, x in per_dispatch(rows, name) if "k_lambda" in k]
        if v:
            res[key + "_per_launch"] = sum(v) / len(v)
            res["launches"] = len(v)
    w = res.get("waves_per_launch") or 10000.0
    for key in ("valu", "salu", "lds"):
        if key + "_per_launch" in res:
            res[key + "_insts_per_query"] = round(res[key + "_per_launch"] / w, 1)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "hist":
        hist(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 1)
    else:
        lam(sys.argv[2], sys.argv[3])
