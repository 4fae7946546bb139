"""What a draw of the --subsample sample costs, for ANY build of the device library (ctypes only, so a
library of an earlier commit loads too): the same seven C-ABI calls, qr_residual_compute timed with and
without a sample.  Round 6 A/B: the hipCUB sort of (key, document) pairs (built from commit 9be0406 into
quickrank_amd/lib/libqr_hip_sortsample.so) against the radix select.

    python scripts/sample_draw_ab.py LIB [LIB ...]

The sort-based library of the A/B (not kept in the tree):
    mkdir /tmp/old && git archive 9be0406 quickrank_amd/csrc include | tar -x -C /tmp/old && cd /tmp/old
    for f in qr_api k_bins k_lambda k_tree k_score k_sample k_wide k_exact; do
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c -o $f.o quickrank_amd/csrc/$f.hip; done
    hipcc --offload-arch=gfx950 -shared -fPIC -o .../quickrank_amd/lib/libqr_hip_sortsample.so *.o"""
import ctypes as C
import sys
import time

import numpy as np


def run(path):
    L = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    L.qr_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.qr_ctx_destroy.argtypes = [vp]
    L.qr_ctx_destroy.restype = None
    L.qr_last_error.argtypes = [vp]
    L.qr_last_error.restype = C.c_char_p
    L.qr_dataset_upload.argtypes = [vp, vp, sz, sz, vp, vp, sz]
    L.qr_bins_build.argtypes = [vp, sz, vp, vp]
    L.qr_subsample_set.argtypes = [vp, C.c_float, C.c_uint64]
    L.qr_residual_compute.argtypes = [vp]
    L.qr_synchronize.argtypes = [vp]

    def ck(h, rc):
        if rc:
            raise RuntimeError(L.qr_last_error(h).decode())

    for N in (1_000_000, 8_000_000):
        rng = np.random.default_rng(0)
        F = 8
        x = rng.random((N, F), dtype=np.float32)
        labels = rng.integers(0, 5, N).astype(np.float32)
        qoff = np.arange(0, N + 1, 100, dtype=np.uint64)
        h = vp()
        ck(h, L.qr_ctx_create(0, C.byref(h)))
        ck(h, L.qr_dataset_upload(h, x.ctypes.data, N, F, labels.ctypes.data, qoff.ctypes.data, len(qoff) - 1))
        thr = np.empty((F, 256), np.float32)
        ts = np.empty(F, np.uint32)
        ck(h, L.qr_bins_build(h, 16, thr.ctypes.data, ts.ctypes.data))

        def cost(reps=50):
            ck(h, L.qr_residual_compute(h))
            ck(h, L.qr_synchronize(h))
            t0 = time.perf_counter()
            for _ in range(reps):
                ck(h, L.qr_residual_compute(h))
            ck(h, L.qr_synchronize(h))
            return (time.perf_counter() - t0) / reps * 1e6

        off = cost()
        for frac in (0.5, 0.1):
            ck(h, L.qr_subsample_set(h, frac, 1))
            on = cost()
            print(f"{path.split('/')[-1]}: N={N} k={frac}N: qr_residual_compute {off:.1f} us without a sample, {on:.1f} us "
                  f"with the draw -> the draw ~{on - off:.1f} us", flush=True)
        L.qr_ctx_destroy(h)


for p in sys.argv[1:]:
    run(p)
