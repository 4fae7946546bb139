"""Two REAL processes on one GPU: each owns a device context and drives the
multi-GPU protocols of quickrank_amd/dist.py through torch.distributed.  RCCL refuses
two ranks on one device, so the transport here is gloo on CUDA tensors (staged
through the host); the contexts, the exchange buffers and the drivers are the real
ones.  The trees must equal the single-context run."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q, layout):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.init()
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import quickrank_amd as qr
    from datagen import make_dataset
    from quickrank_amd.dist import DocShardedTrainer, ShardedTreeFitter, gather_thresholds
    x, labels, qoff = make_dataset(nq=90, docs_per_query=40, F=70, seed=29, adversarial=True)
    N, Q = len(labels), len(qoff) - 1
    stream = torch.cuda.current_stream().cuda_stream
    ok = True
    vx, vl, vq = make_dataset(nq=30, docs_per_query=25, F=70, seed=30)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.upload_valid(vx, vl, vq)
    single.build_bins(64)
    single.reset_scores()
    if layout == "docs":
        cut = [0, 37, Q][rank:rank + 2]
        d0, d1 = int(qoff[cut[0]]), int(qoff[cut[1]])
        c = qr.Context(0, rank=rank, world=world, stream=stream, doc_shard=(N, Q))
        c.upload(x[d0:d1], labels[d0:d1], qoff[cut[0]:cut[1] + 1] - qoff[cut[0]])
        vcut = [0, 11, len(vq) - 1][rank:rank + 2]                 # the validation set is sharded too
        v0, v1 = int(vq[vcut[0]]), int(vq[vcut[1]])
        c.upload_valid(vx[v0:v1], vl[v0:v1], vq[vcut[0]:vcut[1] + 1] - vq[vcut[0]])
        c.build_bins_with(*gather_thresholds(c, 64))
        c.reset_scores()
        tr = DocShardedTrainer(c)
        for it in range(4):
            single.compute_lambdas("NDCG", 10)
            want = single.fit_tree(10, 2, True)
            single.update_scores(0.1)
            tr.compute_lambdas("NDCG", 10)
            got = tr.fit_tree(10, 2, True)
            c.update_scores(0.1)
            for k in ("feature", "thr_id", "left", "right", "nsamples", "threshold"):
                ok = ok and np.array_equal(got[k], want[k])
            ok = ok and np.allclose(got["value"], want["value"], rtol=1e-11, atol=1e-14)
            ok = ok and abs(c.metric_last() - single.metric_last()) < 1e-12
            # training / validation metric over all ranks' queries
            ok = ok and abs(tr.metric_eval(0) - single.metric_eval(0)) < 1e-12
            ok = ok and abs(tr.metric_eval(1) - single.metric_eval(1)) < 1e-12
        ok = ok and np.allclose(c.get_scores(), single.get_scores()[d0:d1], rtol=1e-10, atol=1e-13)
        ok = ok and np.allclose(c.get_valid_scores(), single.get_valid_scores()[v0:v1], rtol=1e-10, atol=1e-13)
    else:
        c = qr.Context(0, rank=rank, world=world, stream=stream)
        c.upload(x, labels, qoff)
        c.build_bins(64)
        c.reset_scores()
        fit = ShardedTreeFitter(c)
        for it in range(4):
            single.compute_lambdas("NDCG", 10)
            want = single.fit_tree(10, 2, True)
            single.update_scores(0.1)
            c.compute_lambdas("NDCG", 10)
            got = fit.fit_tree(c, 10, 2, True)
            c.update_scores(0.1)
            for k in want.dtype.names:
                ok = ok and np.array_equal(got[k], want[k])
        ok = ok and np.array_equal(c.get_scores(), single.get_scores())
    c.close()
    single.close()
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("layout", ["docs", "features"])
def test_two_processes_one_gpu(layout):
    import torch.multiprocessing as mp
    from quickrank_amd import build
    build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + (os.getpid() + (7 if layout == "docs" else 0)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, layout)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}
