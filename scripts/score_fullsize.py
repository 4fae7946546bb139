"""BASELINE.json config 5 at full size: 10,000 trees x 64 leaves over 10M docs x 200
features (8 GB of f32 features generated ON the device, seed 43), through
qr_ensemble_score_device.  Checks a sample of documents bit for bit against a numpy tree walk."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
torch.cuda.init()
from quickrank_amd._capi import Context
from score_bench import make_model, numpy_score

ap = argparse.ArgumentParser()
ap.add_argument("--trees", type=int, default=10000)
ap.add_argument("--docs", type=int, default=10_000_000)
ap.add_argument("--features", type=int, default=200)
ap.add_argument("--check", type=int, default=512)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--ragged", action="store_true",
                help="leaf-wise shaped trees (a random leaf is split until 64: depth ~8-14) instead of complete depth-6 ones")
a = ap.parse_args()
rng = np.random.default_rng(43)
nodes, w = make_model(a.trees, 6, a.features, rng)
depth_stats = None
if a.ragged:
    from score_bench import make_leafwise_model
    nodes, w, depth_stats = make_leafwise_model(a.trees, 64, a.features, rng)
g = torch.Generator(device="cuda")
g.manual_seed(43)
x = torch.rand((a.docs, a.features), generator=g, device="cuda", dtype=torch.float32)
out = torch.empty(a.docs, device="cuda", dtype=torch.float64)
c = Context(0, stream=torch.cuda.current_stream().cuda_stream)
c.upload_ensemble(nodes, w)
L = c.L
def run():
    c._ck(L.qr_ensemble_score_device(c.h, C.c_void_p(x.data_ptr()), a.docs, a.features,
                                     C.c_void_p(out.data_ptr())))
run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
res = {"workload": f"{a.trees} trees x 64 leaves, {a.docs} docs x {a.features} features (device-resident)",
       "ms": ms, "docs_per_s": a.docs / ms * 1e3, "node_visits_per_s": a.docs * a.trees * 6 / ms * 1e3,
       "hbm_alg_GBps": (a.docs * a.features * 4 + a.docs * 8) / ms / 1e6}
if depth_stats:
    res["shape"] = depth_stats
if a.check:
    idx = torch.linspace(0, a.docs - 1, a.check, device="cuda").long()
    want = numpy_score(nodes, w, x[idx].cpu().numpy())
    res["bit_exact_vs_numpy_walk_docs"] = int(a.check) if np.array_equal(out[idx].cpu().numpy(), want) else 0
print(json.dumps(res))
