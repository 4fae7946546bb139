"""Round 6: ONE configuration of the randomised sweep, its FIRST tree only, over and
over in P processes side by side on the one GPU, with everything read back when the device's tree
differs from the oracle's (TEST TOOL, GPU box):

    python tests/tools/repro_first_tree.py SEED INDEX [--procs P] [--iters K] [--torch] [--no-drain] [--jitter]

--jitter: the library built with -DQR_WG_JITTER (quickrank_amd/lib/libqr_jitter.so: one workgroup in
eight of every growth launch starts ~30 us late -- the inter-workgroup race stress of k_tree.hip).

The hunt of scripts/r06_hunt.sh met the intermittent mismatch of rounds 4-5 ~1 time in 50 runs of
config [2] / [248] of seed 0 once EIGHT processes shared the GPU (the round-5 hunts ran one at a
time: 3 events in ~6000 configurations) -- both at tree 0 of a MART run, one of them with the
DEVICE's records inconsistent with the device's own tree (a child's count 5583, the documents its
own split sends there 5603).  This tool stays on that first tree: a fresh context per iteration,
Mart.learn's own call sequence (upload, bins, residuals / lambdas, the batched tree fit), the
oracle's tree computed ONCE per process.  On a mismatch it reads back: the bin map (against the
oracle's, bit for bit), the pseudo-responses, every node's document list against the set the device's
own recorded splits send there, the node histograms' totals -- and fits the tree again on the same
context (does the context repeat it?), then on a fresh one."""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
BODY = r"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, '..')); sys.path.insert(0, os.path.join(%r, '..', '..'))
if not os.environ.get('QR_NO_TORCH'):
    import torch; torch.cuda.init()
import oracle
from datagen import make_dataset
from fuzz_parity import draw_config
from parity_util import assert_tree_parity
from quickrank_amd import Context
seed, index, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
oracle.build(ref=False)
rng = np.random.default_rng(seed)
for i in range(index + 1):
    algo, kw, x, labels, qoff, F, nthr, minls, adv = draw_config(rng, make_dataset)
lam_algo = algo.endswith('LAMBDAMART')
obl = algo.startswith('OBV')
N = len(labels)
print('config', index, algo, N, 'docs', F, 'features', kw, 'adv', adv, flush=True)
tr = oracle.Trainer(x, nthr)
if lam_algo:
    pseudo, weights = oracle.lambdas(labels, np.zeros(N), qoff)[:2]
else:
    pseudo, weights = labels.astype(np.float64), None
ot = tr.fit_tree(pseudo, nleaves=kw.get('nleaves', 10), minls=minls, oblivious_depth=kw.get('depth') if obl else None)
tr.update_output(ot, pseudo, weights)
want = ot['nodes']
crc = (zlib.crc32(x), zlib.crc32(labels), zlib.crc32(qoff))


def fit(c):
    c.reset_scores()
    if lam_algo:
        c.compute_lambdas('NDCG', 10)
    else:
        c.compute_residuals()
    return c.fit_oblivious(kw['depth'], minls, lam_algo) if obl else c.fit_tree(kw['nleaves'], minls, lam_algo)


def same(nodes):
    try:
        assert_tree_parity(tr.stmap, want, nodes, tie_max_docs=1 << 30)
        return None
    except AssertionError as e:
        return e.args


def walk_sets(nodes):
    # the documents the device's OWN recorded splits send to every node, on the oracle's bin map
    sets = {0: np.arange(N)}
    stack = [0]
    while stack:
        k = stack.pop()
        nd = nodes[k]
        if nd['feature'] < 0:
            continue
        d = sets[k]
        go = tr.stmap[nd['feature'], d] <= nd['thr_id']
        sets[int(nd['left'])], sets[int(nd['right'])] = d[go], d[~go]
        stack += [int(nd['left']), int(nd['right'])]
    return sets


bad = 0
for it in range(iters):
    c = Context(0)
    c.upload(x, labels, qoff)
    c.build_bins(nthr)
    nodes = fit(c)
    err = same(nodes)
    if err is None:
        c.close()
        continue
    bad += 1
    print('ITER', it, 'MISMATCH', err, flush=True)
    print(' host inputs unchanged:', (zlib.crc32(x), zlib.crc32(labels), zlib.crc32(qoff)) == crc, flush=True)
    bins = c.read_bins()
    wrong = np.argwhere(bins.T.astype(np.uint32) != tr.stmap)
    print(' device bin map != oracle stmap at', len(wrong), 'cells', wrong[:8].tolist(), flush=True)
    fm = c.read_bins_fm()
    wfm = np.argwhere(fm != bins)
    print(' feature-major copy != block rows at', len(wfm), 'cells', wfm[:8].tolist(),
          'features', np.unique(wfm[:, 1])[:12].tolist() if len(wfm) else [], flush=True)
    dl, dw = c.get_pseudo()
    print(' pseudo-responses differing from the expected ones:', int(np.count_nonzero(~np.isclose(dl, pseudo, rtol=1e-11, atol=1e-14))), flush=True)
    sets = walk_sets(nodes)
    for k in range(len(nodes)):
        nd = nodes[k]
        mark = ''
        try:
            ids = np.sort(c.node_samples(k).astype(np.int64))
        except Exception as e:
            ids = None
            mark = ' (node_samples: %%r)' %% (e,)
        exp = sets.get(k)
        line = ' node %%d: feature %%d thr_id %%d left %%d right %%d nsamples %%d value %%r dev %%r' %% (
            k, nd['feature'], nd['thr_id'], nd['left'], nd['right'], nd['nsamples'], float(nd['value']), float(nd['deviance']))
        if exp is not None and ids is not None:
            only_dev = np.setdiff1d(ids, exp)
            only_walk = np.setdiff1d(exp, ids)
            line += ' | list %%d walked %%d; in the list only %%d %%s, walked only %%d %%s' %% (
                len(ids), len(exp), len(only_dev), only_dev[:6].tolist(), len(only_walk), only_walk[:6].tolist())
            if len(only_dev) and nd['feature'] < 0 and k > 0:
                # the parent's test on the documents that stand in the wrong child
                par = [p for p in range(len(nodes)) if nodes[p]['feature'] >= 0 and k in (int(nodes[p]['left']), int(nodes[p]['right']))]
                if par:
                    p = par[0]
                    f, t = int(nodes[p]['feature']), int(nodes[p]['thr_id'])
                    line += ' | parent %%d tests feature %%d <= %%d; their bins (oracle) %%s (device) %%s' %% (
                        p, f, t, tr.stmap[f, only_dev[:6]].tolist(), bins[only_dev[:6], f].tolist())
        print(line + mark, flush=True)
    try:
        print(' split log:', [(int(s['feature']), int(s['thr_id']), int(s['lcount']), int(s['rcount']), float(s['score'])) for s in c.split_log()], flush=True)
    except Exception as e:
        print(' split log:', repr(e), flush=True)
    again = fit(c)
    print(' the SAME context, fitted again:', 'the oracle\'s tree' if same(again) is None else ('differs again: %%r' %% (same(again),)),
          '| equal to the first attempt:', bool(np.array_equal(again, nodes)), flush=True)
    c.close()
    c2 = Context(0); c2.upload(x, labels, qoff); c2.build_bins(nthr)
    print(' a FRESH context:', 'the oracle\'s tree' if same(fit(c2)) is None else 'differs', flush=True)
    c2.close()
from quickrank_amd import _capi
print('done: %%d of %%d iterations differed; re-reads of polled read-backs: %%d' %% (bad, iters, _capi.READBACK_RETRIES), flush=True)
sys.exit(1 if bad else 0)
""" % (HERE, HERE, HERE)


def main():
    seed, index = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[sys.argv.index("--procs") + 1]) if "--procs" in sys.argv else 8
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 100
    env = dict(os.environ)
    if "--torch" not in sys.argv:
        env["QR_NO_TORCH"] = "1"
    if "--no-drain" not in sys.argv:
        env["QR_DEBUG"] = "1"
    if "--jitter" in sys.argv:
        env["QR_HIP_LIB"] = os.path.join(HERE, "..", "..", "quickrank_amd", "lib", "libqr_jitter.so")
        assert os.path.exists(env["QR_HIP_LIB"]), ("build it first: QR_HIP_LIB=.../libqr_jitter.so "
                                                    "QR_HIP_EXTRA_FLAGS=-DQR_WG_JITTER python -m quickrank_amd.build")
    env.setdefault("OMP_NUM_THREADS", "2")
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, "-X", "faulthandler", "-c", BODY, str(seed), str(index), str(iters)], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(procs)]
    bad = 0
    for k, p in enumerate(ps):
        out = p.communicate()[0]
        lines = [l for l in out.splitlines() if not l.startswith("/opt/amdgpu")]
        if p.returncode != 0:
            bad += 1
            print(f"--- process {k}: rc {p.returncode}\n" + "\n".join(lines)[-12000:], flush=True)
        elif k == 0:
            print(f"--- process 0:\n" + "\n".join(lines)[-600:], flush=True)
    print(f"seed {seed} config {index}: {procs} processes x {iters} iterations, {bad} processes with a mismatch, "
          f"{time.time() - t0:.0f} s", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
