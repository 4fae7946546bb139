"""k_lambda under the microscope (TEST TOOL, GPU box): per-section cycle counts of one
wave (build with -DQR_LAMBDA_TIMING: the kernel prints them for queries 0 and 5000, and those of more than 1100 documents) and
the kernel's duration per boosting iteration (HIP events around qr_lambda_compute) as
the scores lose their ties.   python scripts/lambda_timing.py [iterations]"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
import quickrank_amd.build as b
timing = os.path.join(b.LIBDIR, "libqr_timing.so")
if os.environ.get("QR_LAMBDA_SECTIONS"):
  if not os.path.exists(timing) or os.path.getmtime(timing) < max(os.path.getmtime(os.path.join(b.CSRC, s_)) for s_ in b.SOURCES):
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.FLAGS + ["-DQR_LAMBDA_TIMING", "-o", timing] +
                          [os.path.join(b.CSRC, s) for s in b.SOURCES])
  b.LIB = timing
import quickrank_amd._capi as capi
from bench import synth
if os.environ.get("QR_LT_MSLR"):  # the MSLR-shaped stand-in: the kernel also prints queries of more than 1100 documents
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import make_mslr_like
    x, labels, qoff = make_mslr_like()
else:
    x, labels, qoff = synth(10000, 100, 136)
NQ = len(qoff) - 1
c = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
c.upload(x, labels, qoff); c.build_bins(255); c.reset_scores()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for it in range(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    c.compute_lambdas("NDCG", 10)
    e1.record()
    torch.cuda.synchronize()
    if it in (0, 1, 2, 3, 5, 10, 20, 30, 40, 50, 59) or it == n - 1:
        s = c.get_scores()
        tied = np.mean([(np.diff(np.sort(s[qoff[q]:qoff[q + 1]])) == 0).any() for q in range(0, NQ, 7)])
        print(f"iteration {it}: lambda + prep {e0.elapsed_time(e1) * 1e3:.1f} us, queries with a tied pair {tied:.3f}",
              flush=True)
    c.fit_tree(10, 1, True); c.update_scores(0.1)
