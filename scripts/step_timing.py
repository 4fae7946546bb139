"""Per-section cycle counts of the growth step's control + partition launch (TEST TOOL, GPU
box; the library is built here with -DQR_STEP_TIMING, the kernels print).
python scripts/step_timing.py"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import quickrank_amd.build as b
lib = os.path.join(b.LIBDIR, os.environ.get("QR_TIMING_LIB", "libqr_steptiming.so"))
if not os.path.exists(lib):
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.FLAGS + ["-DQR_STEP_TIMING", "-o", lib] +
                          [os.path.join(b.CSRC, s) for s in b.SOURCES])
b.LIB = lib
import quickrank_amd._capi as capi
from bench import synth
x, labels, qoff = synth(10000, 100, 136)
c = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
c.upload(x, labels, qoff); c.build_bins(255); c.reset_scores()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    print(f"--- iteration {it}", flush=True)
    c.compute_lambdas("NDCG", 10); c.fit_tree(10, 1, True); c.update_scores(0.1)
    torch.cuda.synchronize()
