"""Builds the gfx950 device library in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.environ.get("QR_HIP_LIB") or os.path.join(LIBDIR, "libqr_hip.so")
SOURCES = ["qr_api.hip", "k_bins.hip", "k_lambda.hip", "k_tree.hip", "k_score.hip", "k_sample.hip",
           "k_wide.hip", "k_exact.hip"]
# a measurement aid, NOT part of the product library: the bare LDS atomic rate bench.py prices the
# root histogram launch against (qr_prof_lds_atomic loads it from the product library's directory)
UBENCH_SRC, UBENCH_LIB = "k_ubench.hip", os.path.join(LIBDIR, "libqr_ubench.so")
HEADERS = [os.path.join(CSRC, "qr_internal.h"), os.path.join(CSRC, "qr_wave.h"), os.path.join(CSRC, "qr_dev.h"),
           os.path.join(HERE, "..", "include", "qr_hip.h")]
# -ffp-contract=off: the reference's arithmetic is separate multiply/add
# (SURVEY.md section 7 hard part 8); hipcc contracts by default.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


OBJDIR = os.path.join(LIBDIR, "obj")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _stale():
    return _newer(LIB, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


def build(force=False, verbose=False):
    """One object per .hip source (compiled side by side, only the stale ones), then the link:
    a change to one kernel file costs that file's compile, not all eight."""
    if not force and not _stale():
        build_ubench(verbose=verbose)
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJDIR, exist_ok=True)
    # (QR_HIP_EXTRA_FLAGS + QR_HIP_LIB: experiment builds next to the product library -- their
    # objects are kept apart, keyed by the library's name)
    extra = os.environ.get("QR_HIP_EXTRA_FLAGS", "").split()
    tag = os.path.splitext(os.path.basename(LIB))[0]
    cflags = [f for f in FLAGS if f != "-shared"] + extra
    stamp = os.path.join(OBJDIR, tag + ".flags")
    flags_changed = not os.path.exists(stamp) or open(stamp).read() != " ".join(cflags)

    def compile_one(src):
        obj = os.path.join(OBJDIR, tag + "." + os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if force or flags_changed or _newer(obj, [path] + HEADERS):
            cmd = [hipcc] + cflags + ["-c", "-o", obj, path]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    with open(stamp, "w") as f:
        f.write(" ".join(cflags))
    # (QR_HIP_EXTRA_LDFLAGS: e.g. the host-side AddressSanitizer flavour of tests/tools/abort_hunt.py --asan)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"] + \
        os.environ.get("QR_HIP_EXTRA_LDFLAGS", "").split()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    build_ubench(verbose=verbose)
    return LIB


def build_ubench(force=False, verbose=False):
    src = os.path.join(CSRC, UBENCH_SRC)
    if force or _newer(UBENCH_LIB, [src]):
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + ["-o", UBENCH_LIB, src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return UBENCH_LIB


HOST = os.path.join(HERE, "host")
BINDIR = os.path.join(HERE, "bin")
HOST_SRCS = ["svml.cc", "xml.cc", "mart.cc", "mart_multi.cc", "codegen.cc"]
HOST_LIB = os.path.join(LIBDIR, "libqr_host.so")


def build_host(force=False, verbose=False):
    """C++ host mirror (Mart/LambdaMart, SVMLight, XML, code generators) + the quicklearn/quickscore CLIs."""
    outs = [HOST_LIB, os.path.join(BINDIR, "quicklearn"), os.path.join(BINDIR, "quickscore")]
    srcs = [os.path.join(HOST, f) for f in os.listdir(HOST)]
    if not force and all(os.path.exists(o) for o in outs) and \
            min(os.path.getmtime(o) for o in outs) > max(os.path.getmtime(s) for s in srcs + [LIB]):
        return outs
    os.makedirs(BINDIR, exist_ok=True)
    cxx = os.environ.get("CXX", "g++")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    # (mart_multi.cc calls RCCL: its header needs HIP's types, which g++ gets with the
    # platform macro HIP itself defines under hipcc)
    common = [cxx, "-std=c++17", "-O2", "-fPIC", "-fopenmp", "-pthread", "-Wall", "-Wno-unused-result",
              "-Wno-deprecated-declarations", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include")]
    core = [os.path.join(HOST, f) for f in HOST_SRCS]
    link = ["-L" + LIBDIR, "-lqr_hip", "-L" + os.path.join(rocm, "lib"), "-lrccl",
            "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    cmds = [common + ["-shared", "-o", HOST_LIB] + core + [os.path.join(HOST, "host_capi.cc")] + link,
            common + ["-o", outs[1]] + core + [os.path.join(HOST, "quicklearn.cc")] + link,
            common + ["-o", outs[2]] + core + [os.path.join(HOST, "quickscore.cc")] + link]
    for c in cmds:
        if verbose:
            print(" ".join(c), file=sys.stderr)
        subprocess.check_call(c)
    return outs


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
