"""The headline workload (1M x 136, 10 leaves) and the MSLR-shaped stand-in through ctypes alone --
no torch in the process, i.e. on the /opt/rocm runtime libqr_hip.so links against -- next to the
same loop with torch's bundled HIP runtime mapped first (what bench.py and pytest run on).
TEST TOOL, GPU box:  python scripts/notorch_bench.py [iterations]   (QR_NO_TORCH=1 for the first mode)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synth, synth_mslr
from quickrank_amd._capi import Context
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def run(x, labels, qoff, what):
    c = Context(0); c.upload(x, labels, qoff); c.build_bins(255); c.reset_scores()
    pending = [False]

    def step():
        c.compute_lambdas("NDCG", 10)
        if pending[0]: c.tree_nodes()
        c.fit_tree(10, 1, True, read=False)
        c.update_scores(0.1)
        pending[0] = True
        c.metric_last()
    for _ in range(6): step()
    c.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step()
    c.tree_nodes(); c.synchronize(); dt = (time.perf_counter() - t0) / K
    maps = open('/proc/self/maps').read()
    rt = sorted({l.split()[-1].split('/')[-1] for l in maps.splitlines() if 'libamdhip64' in l})
    print(f"{what}: {dt * 1e3:.4f} ms/iter over {K} iterations, ndcg {c.metric_last():.6f}; torch loaded: {'torch' in sys.modules}; {rt}", flush=True)
    c.close()


run(*synth(10000, 100, 136), "1M x 136 uniform")
run(*synth_mslr(F=136), "MSLR-shaped")
