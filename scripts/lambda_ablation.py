"""k_lambda ablation (TEST TOOL, GPU box): the launch's duration when every wave leaves
after section k (libraries built with -DQR_LAMBDA_STOP=k by `python scripts/lambda_ablation.py
build` on the CPU box; 1 load, 2 counting rank, 3 tie sort, 4 rank + metric, 5 pair sweep,
7 = the whole kernel).  Scores come from a saved run of the full kernel, so every variant
ranks the same queries.   python scripts/lambda_ablation.py [run]"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import quickrank_amd.build as b
STOPS = (1, 2, 8, 3, 4, 5, 7)
lib = lambda k: os.path.join(b.LIBDIR, f"libqr_stop{k}.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    ps = [subprocess.Popen([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.FLAGS + [f"-DQR_LAMBDA_STOP={k}", "-o", lib(k)] +
                           [os.path.join(b.CSRC, s) for s in b.SOURCES]) for k in STOPS]
    sys.exit(max(p.wait() for p in ps))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import numpy as np, torch
    torch.cuda.init()
    b.LIB = lib(int(sys.argv[2]))
    import quickrank_amd._capi as capi
    from bench import synth
    if os.environ.get("QR_ABL_MSLR"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from datagen import make_mslr_like
        x, labels, qoff = make_mslr_like()
    else:
        x, labels, qoff = synth(10000, 100, 136)
    c = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    c.upload(x, labels, qoff); c.build_bins(255); c.reset_scores()
    for name in ("s0", "s30", "s90"):
        sc = np.load(f"/tmp/qr_abl_{name}.npy") if name != "s0" else np.zeros(len(labels))
        c.set_scores(sc)
        ts = []
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); c.compute_lambdas("NDCG", 10); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"stop {sys.argv[2]} scores {name}: lambda + prep {np.median(ts[2:]):.1f} us", flush=True)
    sys.exit(0)
# driver: make the score snapshots with the product library, then one process per variant
import numpy as np, torch
torch.cuda.init()
import quickrank_amd._capi as capi
from bench import synth
if os.environ.get("QR_ABL_MSLR"):     # the MSLR-shaped stand-in (ragged queries of 1..1146 documents)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datagen import make_mslr_like
    x, labels, qoff = make_mslr_like()
else:
    x, labels, qoff = synth(10000, 100, 136)
c = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
c.upload(x, labels, qoff); c.build_bins(255); c.reset_scores()
for it in range(90):
    c.compute_lambdas("NDCG", 10); c.fit_tree(10, 1, True); c.update_scores(0.1)
    if it + 1 in (30, 90):
        s = c.get_scores(); np.save(f"/tmp/qr_abl_s{it + 1}.npy", s)
        tied = float('nan') if os.environ.get('QR_ABL_MSLR') else (np.diff(np.sort(s.reshape(10000, 100), axis=1), axis=1) == 0).any(axis=1).mean()
        print(f"after {it + 1} trees: queries with a tied pair {tied:.3f}", flush=True)
del c
for k in STOPS:
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "one", str(k)])
