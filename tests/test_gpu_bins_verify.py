"""Verify-after-write of the resident bin map (round 6; profiles/r06_hunt.md): the map RTRootHistogram
builds once (rtnode_histogram.cc:227-253) is read by every launch of every tree, and the r06 hunt saw the
platform drop one workgroup in eight of exactly its two launches under GPU oversubscription.
qr_bins_build checks both copies against a recomputation on the device and builds them again when a
cell differs; these tests drive the check and the rebuild through the C-ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qr():
    import quickrank_amd
    from quickrank_amd import build
    build.build()
    return quickrank_amd


def _ctx(qr, nthr=32, F=50, seed=3):
    from datagen import make_dataset
    x, labels, qoff = make_dataset(nq=60, docs_per_query=40, F=F, seed=seed)
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    c.build_bins(nthr)
    return c, x, labels, qoff


def test_a_fresh_map_verifies(qr):
    c, *_ = _ctx(qr)
    assert c.verify_bins() == (0, 0)
    c.close()


@pytest.mark.parametrize("which", [0, 1], ids=["block_rows", "feature_major"])
def test_lost_stores_are_seen(qr, which):
    """Rows zeroed behind the kernels' backs -- what a workgroup's lost stores look like -- are counted
    in the copy they were lost from, and only there."""
    c, x, *_ = _ctx(qr)
    bins = c.read_bins()
    c.debug_clobber_bins(which, 16, 8)
    want = int(np.count_nonzero(bins[16:24]))          # a zeroed cell differs wherever the true bin is not 0
    got = c.verify_bins()
    assert got[which] == want > 0 and got[1 - which] == 0
    c.close()


def test_a_build_that_loses_stores_is_built_again(qr, capfd, oracle_lib):
    """One build that loses eight documents' rows: qr_bins_build notices, says so on stderr, builds the
    map again and hands out the right one."""
    from datagen import make_dataset
    x, labels, qoff = make_dataset(nq=60, docs_per_query=40, F=50, seed=3)
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    c.debug_clobber_bins(2, 1, 0)
    c.build_bins(32)
    err = capfd.readouterr().err
    assert "does not hold what the binning kernel stored" in err and "building it again" in err
    tr = oracle_lib.Trainer(x, 32)
    assert np.array_equal(c.read_bins().T.astype(np.uint32), tr.stmap)
    assert c.verify_bins() == (0, 0)
    c.close()


def test_a_map_that_never_holds_is_an_error(qr, capfd):
    from datagen import make_dataset
    x, labels, qoff = make_dataset(nq=60, docs_per_query=40, F=50, seed=3)
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    c.debug_clobber_bins(2, 3, 0)
    with pytest.raises(qr.QrError, match="three times in a row"):
        c.build_bins(32)
    capfd.readouterr()
    c.close()


@pytest.mark.parametrize("algo", ["leafwise", "oblivious"])
def test_a_tree_whose_counts_do_not_add_up_is_refused(qr, algo):
    """The other half of the defence (round 6): the hunt's run 7 trained on a root histogram 4,096
    documents short -- bin map good, the partial slot of one workgroup never stored.  qr_tree_nodes
    checks the records' arithmetic (the root holds the documents the tree was grown on, an internal node
    as many as its children; the counts come from different launches) before it hands them out."""
    c, x, labels, qoff = _ctx(qr)
    fit = (lambda: c.fit_oblivious(3, 2, False)) if algo == "oblivious" else (lambda: c.fit_tree(8, 2, False))
    c.compute_residuals()
    nodes = fit()                                   # an honest tree passes, and says so itself
    assert nodes[0]["nsamples"] == len(labels)
    inner = np.nonzero(nodes["feature"] >= 0)[0]
    assert all(nodes[int(nodes[i]["left"])]["nsamples"] + nodes[int(nodes[i]["right"])]["nsamples"] == nodes[i]["nsamples"]
               for i in inner)
    c.debug_clobber_bins(3, 4096 if len(labels) > 4096 else 7, 0)
    c.compute_residuals()
    with pytest.raises(qr.QrError, match="do not add up .the root counts"):
        fit()
    c.compute_residuals()
    again = fit()                                   # (the aid is one-shot: the context is fine)
    assert again[0]["nsamples"] == len(labels)
    c.close()
