"""The intermittent mismatch of round 5's hunt -- sweep seed 1, configuration 18 (LAMBDAMART, 5066
documents, 5 features, 64 thresholds, 64 leaves), tree 0, a leaf's value -- taken apart (TEST TOOL,
GPU box): one fresh process per trial, the configuration alone (`--alone`) or behind the
configurations before it, the first tree step by step with the pseudo-responses read back:
    python tests/tools/mismatch_diag.py TRIALS [--alone]
QR_DEBUG=1 (drain + check after every call) is what made it show up once in ~25 processes."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
BODY = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, '..')); sys.path.insert(0, os.path.join(%r, '..', '..'))
import torch; torch.cuda.init()
import oracle
from datagen import make_dataset
from quickrank_amd.trainer import Mart
from quickrank_amd import Context
from parity_util import assert_tree_parity
alone = sys.argv[1] == '1'
if not alone:
    from fuzz_parity import sweep
    sweep(30, 0, verbose=False)          # the hunt's first sweep, as it ran
rng = np.random.default_rng(1)   # (the hunt's SECOND sweep: seed 1)
oracle.build(ref=False)
for i in range(19):
    F = int(rng.choice([5, 9, 16, 17, 40, 64, 65, 136, 200])); nq = int(rng.integers(5, 400))
    dpq = int(rng.choice([1, 3, 16, 17, 40, 100, 250])); nthr = int(rng.choice([2, 8, 16, 64, 255]))
    algo = str(rng.choice(["LAMBDAMART", "MART", "OBVLAMBDAMART", "OBVMART"])); minls = int(rng.choice([1, 1, 2, 5, 20]))
    kw = dict(ntrees=int(rng.integers(2, 6)), shrinkage=0.1, nthresholds=nthr, minls=minls, esr=0)
    if algo.startswith("OBV"): kw["depth"] = int(rng.integers(1, 7))
    else: kw["nleaves"] = int(rng.choice([2, 3, 8, 10, 31, 64]))
    dseed, ragged, adversarial = int(rng.integers(1 << 30)), bool(rng.integers(2)), bool(rng.integers(2))
    x, labels, qoff = make_dataset(nq=nq, docs_per_query=dpq, F=F, seed=dseed, ragged=ragged, adversarial=adversarial)
    if i < 18:
        if not alone:
            m = Mart(algo=algo, **kw).learn(x, labels, qoff); m.ctx.close()
        continue
    sizes = np.diff(qoff.astype(np.int64))
    print('config 18:', algo, len(labels), 'docs', F, 'features', kw, 'queries', len(sizes), 'sizes', sizes.min(), sizes.max(), 'ragged', ragged, 'adv', adversarial, flush=True)
    # Mart.learn's own call sequence (nothing between the lambda pass and the tree), the checks
    # BEHIND each tree: the pseudo-responses the tree was grown on, the oracle's tree on them
    c = Context(0); c.upload(x, labels, qoff); c.build_bins(nthr); c.reset_scores()
    tr = oracle.Trainer(x, nthr)
    failed = False
    for t in range(kw["ntrees"]):
        c.compute_lambdas("NDCG", 10)
        nodes = c.fit_tree(kw["nleaves"], minls, True)
        lam, w = c.get_pseudo()
        sc = c.get_scores()
        olam, ow = oracle.lambdas(labels, sc, qoff, 10, 1)
        bl = np.nonzero(~np.isclose(lam, olam, rtol=1e-10, atol=1e-14))[0]; bw = np.nonzero(~np.isclose(w, ow, rtol=1e-10, atol=1e-14))[0]
        ot = tr.fit_tree(lam, nleaves=kw["nleaves"], minls=minls); tr.update_output(ot, lam, w)
        ok = True
        try:
            assert_tree_parity(tr.stmap, ot["nodes"], nodes, tie_max_docs=1 << 30)
        except AssertionError as e:
            ok = False
            print('TREE', t, 'MISMATCH', e.args, flush=True)
        if len(bl) or len(bw) or not ok:
            failed = True
            print('DIAG tree', t, 'lambdas differing from the oracle\'s on the device\'s scores:', len(bl), 'weights:', len(bw), flush=True)
            for k in range(len(nodes)):
                if nodes[k]['feature'] < 0 and not np.isclose(nodes[k]['value'], ot['nodes'][k]['value'], rtol=1e-9, atol=1e-10):
                    ids = c.node_samples(k).astype(np.int64)
                    hv = lam[ids].sum() / w[ids].sum() if w[ids].sum() >= 2.2e-16 else 0.0
                    print(' leaf', k, 'device value', repr(float(nodes[k]['value'])), 'oracle (on the device\'s lambdas)', repr(float(ot['nodes'][k]['value'])),
                          'n', int(nodes[k]['nsamples']), 'list length', len(ids), 'ids ascending', bool(np.all(np.diff(ids) > 0)),
                          'host sum over the device\'s list:', repr(float(hv)), flush=True)
        c.update_scores(0.1)
    print('trial', 'FAILED' if failed else 'ok', flush=True)
    c.close()
""" % (HERE, HERE, HERE)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
alone = "--alone" in sys.argv
bad = 0
for t in range(trials):
    out = subprocess.run([sys.executable, "-c", BODY, "1" if alone else "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    failed = "trial ok" not in out
    bad += failed
    if failed or t == 0:
        print(f"--- trial {t}{' (alone)' if alone else ''}:\n" + "\n".join(l for l in out.splitlines() if not l.startswith("/opt/amdgpu"))[-3000:], flush=True)
print(f"{trials} trials{' (alone)' if alone else ''}: {bad} failed", flush=True)
