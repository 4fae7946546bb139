"""Tree comparison shared by the GPU parity tests and __graft_entry__.smoke()."""
import numpy as np


def assert_tree_parity(stmap, onodes, gnodes, exact=False, value_rtol=1e-7):
    """Device tree vs oracle tree, walked together from the root.

    At every internal node the device's split must cut the node's documents into
    the same two sets as the oracle's.  The recorded (feature, slot) must be the
    oracle's too, except when another candidate induces exactly the same
    two-set partition (possibly with left/right mirrored): such candidates have
    equal gains in exact arithmetic, the reference picks among them by the
    rounding noise of its f64 summation order (DESIGN.md "Parity"), the device
    takes the lexicographically first.  Leaves must hold the same documents and
    values.  Returns the number of tie-resolved splits; exact=True forbids them.
    """
    ties = 0
    stack = [(0, 0, np.arange(stmap.shape[1]))]
    nleaves = 0
    while stack:
        oi, gi, d = stack.pop()
        o, g = onodes[oi], gnodes[gi]
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
            ties += 1
        gL, gR = (g["right"], g["left"]) if mirrored else (g["left"], g["right"])
        stack.append((int(o["left"]), int(gL), d[ol]))
        stack.append((int(o["right"]), int(gR), d[~ol]))
    assert nleaves == int((onodes["feature"] < 0).sum()) == int((gnodes["feature"] < 0).sum())
    return ties


