"""Tree comparison shared by the GPU parity tests and __graft_entry__.smoke()."""
import numpy as np

# A split may deviate from the oracle's (feature, slot) only where another candidate
# induces exactly the same two-set partition of the node's documents, and only in
# nodes of at most this many documents (DESIGN.md "Parity": such candidates have
# equal gains in exact arithmetic; the reference picks among them by the rounding
# noise of its f64 summation order, the device takes the lexicographically first).
# Larger nodes must match (feature, slot) bit for bit.
TIE_MAX_DOCS = 1000


class Ties(int):
    """Number of tie-resolved splits (an int), plus what the callers need to
    compare the rest of the two trees *modulo* those ties."""
    sizes = ()      # documents in each tie-resolved node
    onodes = ()     # oracle node index of each tie-resolved split
    node_map = {}   # oracle node index -> device node index
    mirrored = ()   # oracle node indices whose device children are swapped


def assert_tree_parity(stmap, onodes, gnodes, exact=False, value_rtol=1e-7,
                       tie_max_docs=TIE_MAX_DOCS, docs=None):
    """Device tree vs oracle tree, walked together from the root.

    At every internal node the device's split must cut the node's documents into
    the same two sets as the oracle's.  The recorded (feature, slot) must be the
    oracle's too, except when another candidate induces exactly the same
    two-set partition (possibly with left/right mirrored) in a node of at most
    `tie_max_docs` documents.  Leaves must hold the same documents and values.
    Returns a `Ties` (an int: the number of tie-resolved splits, with their node
    sizes and the oracle -> device node map); exact=True forbids ties altogether.
    `docs`: the root's document ids (default: every column of stmap).
    """
    sizes, tied, mirrored_nodes = [], [], []
    node_map = {}
    stack = [(0, 0, np.arange(stmap.shape[1]) if docs is None else np.asarray(docs))]
    nleaves = 0
    while stack:
        oi, gi, d = stack.pop()
        o, g = onodes[oi], gnodes[gi]
        node_map[int(oi)] = int(gi)
        assert o["nsamples"] == g["nsamples"] == len(d), (oi, gi)
        assert (o["feature"] < 0) == (g["feature"] < 0), (oi, gi)
        if o["feature"] < 0:
            assert np.isclose(g["value"], o["value"], rtol=value_rtol, atol=1e-10), (oi, gi)
            nleaves += 1
            continue
        ol = stmap[o["feature"], d] <= o["thr_id"]
        if (o["feature"], o["thr_id"]) == (g["feature"], g["thr_id"]):
            assert g["threshold"].view(np.uint32) == o["threshold"].view(np.uint32)
            mirrored = False
        else:
            assert not exact, (oi, o["feature"], o["thr_id"], g["feature"], g["thr_id"])
            gl = stmap[g["feature"], d] <= g["thr_id"]
            if np.array_equal(gl, ol):
                mirrored = False
            elif np.array_equal(gl, ~ol):
                mirrored = True
            else:
                raise AssertionError(("different partition at oracle node", oi))
            assert len(d) <= tie_max_docs, ("equal-partition tie in a node of", len(d), "documents", oi)
            sizes.append(len(d))
            tied.append(int(oi))
            if mirrored:
                mirrored_nodes.append(int(oi))
        gL, gR = (g["right"], g["left"]) if mirrored else (g["left"], g["right"])
        stack.append((int(o["left"]), int(gL), d[ol]))
        stack.append((int(o["right"]), int(gR), d[~ol]))
    assert nleaves == int((onodes["feature"] < 0).sum()) == int((gnodes["feature"] < 0).sum())
    t = Ties(len(sizes))
    t.sizes, t.onodes, t.node_map, t.mirrored = tuple(sizes), tuple(tied), node_map, tuple(mirrored_nodes)
    return t


def assert_split_log_parity(log, olog, ties, score_rtol=1e-9):
    """Split logs in growth order, compared modulo the tie-resolved splits: every
    entry cuts off the same two child sizes with the same gain; (feature, slot) may
    differ in at most `ties` entries (and then only with the counts possibly
    mirrored)."""
    assert len(log) == len(olog)
    lo = np.minimum(log["lcount"], log["rcount"])
    hi = np.maximum(log["lcount"], log["rcount"])
    olo = np.minimum(olog["lcount"], olog["rcount"])
    ohi = np.maximum(olog["lcount"], olog["rcount"])
    same = (log["feature"].astype(np.uint64) == olog["feature"]) & \
           (log["thr_id"].astype(np.uint64) == olog["thr_id"])
    if int(ties) == 0:
        assert same.all()
        assert np.array_equal(log["lcount"], olog["lcount"])
        assert np.array_equal(log["rcount"], olog["rcount"])
        assert np.allclose(log["score"], olog["score"], rtol=score_rtol)
        return
    # A mirrored tie swaps the two children's node indices; the heap keys (the
    # children's deviances) are the same numbers, so only the order of equal keys can
    # change: compare the logs as multisets of (smaller child, larger child).
    got = np.lexsort((log["score"], hi, lo))
    want = np.lexsort((olog["score"], ohi, olo))
    assert np.array_equal(lo[got], olo[want]) and np.array_equal(hi[got], ohi[want])
    assert np.allclose(log["score"][got], olog["score"][want], rtol=score_rtol)
    if not ties.mirrored:  # same growth order: only the tie-resolved entries may name another candidate
        assert int((~same).sum()) <= int(ties)


def assert_same_tree_records(got, want, node_sums_exact=True, where=None):
    """Two device runs of the same tree (e.g. a sharded protocol against the single context).
    Every field bit for bit, except -- `node_sums_exact=False` -- the bookkeeping that comes
    from a node's f64 sums of pseudo-responses (`deviance` of every node, `value` of INTERNAL
    nodes): batched single-GPU growth takes those sums from its histogram pass, the
    one-split-per-step protocols from their partition pass -- two fixed summation orders, equal
    to rounding (1e-11 here).  What leaves the trainer -- structure, thresholds, sample
    counts, leaf outputs -- stays exact."""
    assert len(got) == len(want), where
    leaf = want["feature"] < 0
    for k in want.dtype.names:
        if not node_sums_exact and k in ("value", "deviance"):
            if k == "value":
                assert np.array_equal(got[k][leaf], want[k][leaf]), (where, "leaf outputs")
            assert np.allclose(got[k], want[k], rtol=1e-11, atol=1e-13, equal_nan=True), (where, k)
        elif want[k].dtype.kind == "f":
            assert np.array_equal(got[k], want[k], equal_nan=True), (where, k)
        else:
            assert np.array_equal(got[k], want[k]), (where, k)
