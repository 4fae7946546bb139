"""Per-section cycle counts of k_leaf_sums_doc (TEST TOOL, GPU box; builds the library with
-DQR_LEAF_TIMING, workgroups 0 and 500 print):  python scripts/leaf_timing.py"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import quickrank_amd.build as b
lib = os.path.join(b.LIBDIR, "libqr_leaftiming.so")
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.FLAGS + ["-DQR_LEAF_TIMING", "-o", lib] + [os.path.join(b.CSRC, s) for s in b.SOURCES])
b.LIB = lib
import quickrank_amd._capi as capi
from bench import synth
x, labels, qoff = synth(10000, 100, 136)
c = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
c.upload(x, labels, qoff); c.build_bins(255); c.reset_scores()
for it in range(6):
    c.compute_lambdas("NDCG", 10); c.fit_tree(10, 1, True); c.update_scores(0.1)
    torch.cuda.synchronize()
