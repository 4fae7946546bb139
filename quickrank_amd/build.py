"""Builds the gfx950 device library in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libqr_hip.so")
SOURCES = ["qr_api.hip", "k_bins.hip", "k_lambda.hip", "k_tree.hip", "k_score.hip"]
HEADERS = [os.path.join(CSRC, "qr_internal.h"),
           os.path.join(HERE, "..", "include", "qr_hip.h")]
# -ffp-contract=off: the reference's arithmetic is separate multiply/add
# (SURVEY.md section 7 hard part 8); hipcc contracts by default.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
