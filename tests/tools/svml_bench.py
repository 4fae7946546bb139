#!/usr/bin/env python3
"""SURVEY section 8 row f2 at the measurement bar: the SVMLight reader either side of the hot path.

    python tests/tools/svml_bench.py [--docs 200000] [--features 136] [--repeat 3] [--md profiles/rNN_svml_reader.md]

Writes an MSLR-shaped file (100-document queries, every feature present, %.6f values, a `#docid`
trailer on every tenth line), then times
  * quickrank_amd/host/svml.cc (`qrh_svml_read`): the chunked parallel reader, OMP_NUM_THREADS as set;
  * the same with one thread (what the grammar costs, without the chunking);
  * the reference's own reader (svml.cc:38-161: getline + sscanf, serial), through oracle/_ref --
    only where /root/reference exists (the build container); elsewhere the line is left out.
Arrays of all readers are compared bit for bit before any time is reported.  CPU only.  Lives under
tests/ because it loads oracle/_ref (test infrastructure) for the baseline."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sz = C.c_size_t


def bind(lib, name):
    fn = getattr(lib, name)
    fn.argtypes = [C.c_char_p, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    return fn


def read(fn, path):
    """(both C shims read the file once per call: the sizing call is not timed)"""
    N, F, Q = sz(0), sz(0), sz(0)
    rc = fn(path.encode(), C.byref(N), C.byref(F), C.byref(Q), None, None, None)
    assert rc == 0, rc
    x = np.empty((N.value, F.value), np.float32)
    lab = np.empty(N.value, np.float32)
    qoff = np.empty(Q.value + 1, np.uint64)
    t0 = time.perf_counter()
    rc = fn(path.encode(), C.byref(N), C.byref(F), C.byref(Q), x.ctypes.data, lab.ctypes.data, qoff.ctypes.data)
    assert rc == 0, rc
    return (x, lab, qoff), time.perf_counter() - t0


def write_file(path, docs, F, seed=42):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for a in range(0, docs, 20000):
            n = min(20000, docs - a)
            x = rng.random((n, F), np.float32)
            lab = rng.integers(0, 5, n)
            rows = []
            for i in range(n):
                d = a + i
                feats = " ".join(f"{k + 1}:{v:.6f}" for k, v in enumerate(x[i]))
                rows.append(f"{lab[i]} qid:{d // 100 + 1} {feats}" + (f" #docid = {d}" if d % 10 == 0 else ""))
            f.write("\n".join(rows) + "\n")
    return os.path.getsize(path)


def timed_in_child(which, path, threads, repeat):
    """(a fresh process per configuration: OMP_NUM_THREADS is read once, and the page cache is warm
    for every reader alike -- the file was just written)"""
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests', 'tools')!r})\n"
            f"import svml_bench as B, ctypes as C\n"
            f"fn = B.reader({which!r})\n"
            f"ts = [B.read(fn, {path!r})[1] for _ in range({repeat})]\n"
            f"print(min(ts))\n")
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True)
    return float(out.stdout.strip().splitlines()[-1])


def reader(which):
    if which == "ours":
        from quickrank_amd import build
        build.build_host()
        return bind(C.CDLL(build.HOST_LIB), "qrh_svml_read")
    import oracle
    oracle.build(ref=True)
    return bind(oracle.ref(), "ref_svml_read")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=200000)
    ap.add_argument("--features", type=int, default=136)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--md", default=None)
    a = ap.parse_args()
    have_ref = os.path.isdir("/root/reference/src")
    cores = len(os.sched_getaffinity(0))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "bench.svml")
        nbytes = write_file(path, a.docs, a.features)
        mb = nbytes / 1e6
        ours, _ = read(reader("ours"), path)
        if have_ref:
            theirs, _ = read(reader("ref"), path)
            for u, v in zip(ours, theirs):
                assert u.shape == v.shape and u.tobytes() == v.tobytes(), "readers disagree"
        rows = [("`host/svml.cc`, %d threads" % cores, timed_in_child("ours", path, cores, a.repeat)),
                ("`host/svml.cc`, 1 thread", timed_in_child("ours", path, 1, a.repeat))]
        if have_ref:
            rows.append(("reference `Svml::read_horizontal` (svml.cc:38-161), serial", timed_in_child("ref", path, 1, a.repeat)))
    lines = [f"| reader | seconds (best of {a.repeat}) | MB/s | documents/s |", "|---|---|---|---|"]
    for name, t in rows:
        lines.append(f"| {name} | {t:.3f} | {mb / t:.0f} | {a.docs / t:.3g} |")
    head = (f"{a.docs} documents x {a.features} features in {a.docs // 100} queries, {mb:.0f} MB of text, page cache warm; "
            f"arrays of the readers compared bit for bit first ({'equal' if have_ref else 'reference reader not present here'}).")
    text = head + "\n\n" + "\n".join(lines) + "\n"
    if have_ref:
        text += f"\nspeed-up over the reference's reader: {rows[2][1] / rows[0][1]:.1f}x with {cores} threads, {rows[2][1] / rows[1][1]:.1f}x with one.\n"
    print(text)
    if a.md:
        with open(os.path.join(ROOT, a.md), "w") as f:
            f.write("# SVMLight reader (SURVEY 8 row f2): `tests/tools/svml_bench.py`, build container's host cores\n\n" + text)


if __name__ == "__main__":
    main()
