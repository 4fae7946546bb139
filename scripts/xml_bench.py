#!/usr/bin/env python3
"""SURVEY section 8 row f1 at the measurement bar: loading and saving a model file.

    python scripts/xml_bench.py [--trees 10000] [--leaves 64] [--repeat 3] [--md profiles/rNN_xml_model.md]

Writes a model of config 5's shape (10,000 leaf-wise trees of 64 leaves over 200 features, the
look of `quicklearn --model-out`), then times `Mart::load_model_from_file` (quickscore's and
`--restart-train`'s way in) and `Mart::save` through quickrank_amd/host's C shim, with all host
threads and with one, and checks that load + save reproduces the file byte for byte.  The
reference reads and writes through pugixml, an un-vendored submodule absent from this image, so
there is no reference line here (parity of the number format is unpinned)."""
import argparse
import ctypes as C
import filecmp
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sz = C.c_size_t


def host():
    from quickrank_amd import build
    build.build_host()
    L = C.CDLL(build.HOST_LIB)
    L.qrh_model_write.argtypes = [C.c_char_p, C.c_int, sz, C.c_double, sz, sz, sz, sz, sz, C.c_void_p, sz, sz]
    L.qrh_model_roundtrip_timed.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return L


def child(path, out, threads, repeat):
    code = (f"import sys; sys.path.insert(0, {os.path.join(ROOT, 'scripts')!r}); import xml_bench as B, ctypes as C\n"
            f"L = B.host(); best = [1e9, 1e9]\n"
            f"for _ in range({repeat}):\n"
            f"    a, b = C.c_double(0), C.c_double(0)\n"
            f"    assert L.qrh_model_roundtrip_timed({path!r}.encode(), {out!r}.encode(), C.byref(a), C.byref(b)) == 0\n"
            f"    best = [min(best[0], a.value), min(best[1], b.value)]\n"
            f"print(best[0], best[1])\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS=str(threads)),
                       capture_output=True, text=True, check=True)
    return tuple(float(v) for v in r.stdout.strip().splitlines()[-1].split())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=10000)
    ap.add_argument("--leaves", type=int, default=64)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--md", default=None)
    a = ap.parse_args()
    from score_bench import make_leafwise_model
    nodes = np.ascontiguousarray(make_leafwise_model(a.trees, a.leaves, 200, np.random.default_rng(1))[0])
    cores = len(os.sched_getaffinity(0))
    with tempfile.TemporaryDirectory() as tmp:
        p, q = os.path.join(tmp, "m.xml"), os.path.join(tmp, "again.xml")
        assert host().qrh_model_write(p.encode(), 1, a.trees, 0.1, 255, a.leaves, 1, 0, 0, nodes.ctypes.data,
                                      a.trees, nodes.shape[1]) == 0
        mb = os.path.getsize(p) / 1e6
        rows = [(f"{cores} host threads",) + child(p, q, cores, a.repeat)]
        same = filecmp.cmp(p, q, shallow=False)
        rows.append(("1 host thread",) + child(p, q, 1, a.repeat))
    assert same, "load + save does not reproduce the file"
    text = (f"{a.trees} trees x {a.leaves} leaves ({a.trees * (2 * a.leaves - 1)} nodes), {mb:.0f} MB of XML; "
            f"load + save reproduces the file byte for byte; best of {a.repeat}.\n\n"
            "| | `Mart::load_model_from_file` | MB/s | `Mart::save` | MB/s |\n|---|---|---|---|---|\n")
    for name, ld, sv in rows:
        text += f"| {name} | {ld:.2f} s | {mb / ld:.0f} | {sv:.2f} s | {mb / sv:.0f} |\n"
    print(text)
    if a.md:
        with open(os.path.join(ROOT, a.md), "w") as f:
            f.write("# Model file (SURVEY 8 row f1): `scripts/xml_bench.py`, build container's host cores\n\n" + text)


if __name__ == "__main__":
    main()
