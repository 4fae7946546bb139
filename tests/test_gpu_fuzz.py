"""The randomised parity sweep (tests/tools/fuzz_parity.py) inside `pytest -m gpu`:
300 random configurations of seed 0 -- algorithm, N from 8 to ~50k with ragged /
adversarial columns, F from 5 to 200, 2..255 thresholds, 2..64 leaves or depth
1..6, min leaf support 1..20 -- device trees against the oracle's.

Every run must match as tests/parity_util.py defines it (ties only between
candidates that cut a node of <= 1000 documents into the same two sets), EXCEPT the
runs named below: MART runs (and one first tree of a LambdaMART run, where all scores are
still 0) on tiny / many-leaved sets where the reference decides
a split by the rounding noise of its f64 summation order -- two different
partitions with gains equal in exact arithmetic, or the `deviance > 0` gate of
rt.cc:212 on a node whose residuals are all equal.  The sweep verifies each is
exactly that (child deviances summing to the same total / |deviance| < 1e-9 of the
root's) and cuts the run short there; they are asserted by name so that a new
divergence, or one of these disappearing, fails the test."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))

# configuration index -> what the sweep reported on the round-1 state
# runs of a sweep that may end at a split the reference decides by rounding noise (every one
# classified): the ceiling the extra seeds are held to (rounds 2-3 measured 0.3 - 4.3 % per seed of 300; seeds 1 and 2 here 1.5 and 2.5 %
# per 300-configuration seed, 1.9 % over 12,600 configurations)
DIVERGENCE_CEILING = 0.03

KNOWN_ROUNDING_DECIDED = {58: "MART N=1035 F=200 nthr=16 minls=2 64",
                          67: "MART N=393 F=200 nthr=64 minls=2 64",
                          174: "MART N=23 F=136 nthr=255 minls=2 31",
                          204: "MART N=158 F=65 nthr=64 minls=5 31",
                          214: "LAMBDAMART N=195 F=200 nthr=8 minls=5 31"}


def _assert_fp_tie_is_the_references_rounding(r):
    """a `gain_tie_fp` -- two candidates whose exact gains differ, by less than
    the device's 33-bit fixed-point gradients resolve -- carries its distance in units of the last
    place: of the EXACT gains (rational arithmetic on the oracle's pseudo-responses) and of the
    gains the reference's own f64 sums give (rtnode_histogram.cc:51-69, rt.cc:276-279).  Every
    instance met so far (seeds 0-2: two) is under ONE ulp in exact arithmetic and two ulps apart in
    the reference's f64 -- the reference's pick is its summation order's, and exact arithmetic
    prefers the device's candidate.  More than 4 ulps would be a split the reference resolves
    and the fixed point does not: that fails here, by name."""
    if r["status"] != "gain_tie_fp":
        return
    v = r["ref_view"]
    print(f"gain_tie_fp {r['desc']} tree {r['tree']}: exact gains {v['exact_ulps_apart']:.3g} ulps apart, the "
          f"reference's f64 gains {v['f64_ulps_apart']:.1f} ulps; exact arithmetic prefers the {v['exact_prefers']} candidate")
    assert v["exact_ulps_apart"] <= 4.0, r


def _report_oracle_reruns(res):
    """A configuration judged twice because two runs of the ORACLE on the same inputs gave different
    trees (fuzz_parity.oracle_rerun_differs) is printed with both values -- and FAILS the sweep
    (round 6; rounds 4-5 tolerated one per sweep).  What round 6's hunt found
    (profiles/r06_hunt.md): 144,000 sweep configurations in 400 processes, four at a time on the GPU,
    every device tree compared behind its fit: not one event of either kind; the events of rounds
    4-6 need eight processes and a host oversubscribed eight times over, and are then the DEVICE's
    memory -- the stores of one workgroup in eight of an element-wise kernel missing, GPU page
    faults beside them -- not the checker's."""
    again = [r for r in res if r.get("oracle_reruns")]
    for r in again:
        print("ORACLE NOT REPRODUCIBLE (judged again):", r["desc"], r["oracle_diff"])
    assert len(again) == 0, [r["desc"] for r in again]


def test_fuzz_sweep_seed0():
    from fuzz_parity import sweep
    from parity_util import TIE_MAX_DOCS
    res = sweep(300, 0, verbose=False)
    assert len(res) == 300
    _report_oracle_reruns(res)
    cut = {r["i"]: r for r in res if r["status"] != "ok"}
    assert set(cut) == set(KNOWN_ROUNDING_DECIDED), {i: r["desc"] for i, r in cut.items()}
    for i, r in cut.items():
        # only discrete pseudo-responses tie exactly: MART's residuals, or LambdaMART's
        # first tree (all scores 0: a query's lambdas take a handful of values)
        assert r["desc"].split()[1] == "MART" or r["tree"] == 0, r["desc"]
        assert KNOWN_ROUNDING_DECIDED[i] in r["desc"], r["desc"]
        print("rounding-decided:", r["desc"], r["status"], "tree", r["tree"], "rest:", r.get("rest"))
        _assert_fp_tie_is_the_references_rounding(r)
        # (round 6) the trees behind the cut were verified one by one against what the reference's
        # algorithm builds from the device's own scores (fuzz_parity.verify_rest_causally raises otherwise)
        assert "rest" in r and all(k != "unclassified" for _, k in r["rest"]), (r["desc"], r["rest"])
    sizes = [s for r in res for s in r["tie_sizes"]]
    # (seed 0's ties happen to sit in nodes of <= TIE_MAX_DOCS documents; that is a regression
    # marker for THIS seed, not a property of the design -- see the other seeds' test)
    assert all(s <= TIE_MAX_DOCS for s in sizes)
    # ties are the exception, not the rule: far fewer than one per run
    assert sum(r["ties"] for r in res) <= 150, sum(r["ties"] for r in res)
    print(f"fuzz: {sum(r['ties'] for r in res)} equal-partition ties over 300 runs, largest node "
          f"{max(sizes or [0])} documents; {len(cut)} runs decided by rounding noise")


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_sweep_more_seeds_divergence_rate(seed):
    """two more seeds of the sweep inside `pytest -m gpu`, asserting what is TRUE
    of the fixed-point design instead of a node-size bound that only seed 0 kept: (i) every run ends
    in "ok" or in one of the CLASSIFIED kinds -- an exact gain tie between different partitions
    (priced in exact rational arithmetic on the oracle's pseudo-responses), a gain difference below
    the fixed-point gradients' resolution, a zero-deviance gate, a heap order or a ranking decided
    by rounding (the latter verified causally: the device's tree is what the reference's algorithm
    builds from the device's own scores); anything else raises inside sweep(); (ii) an
    equal-partition tie is a split that cuts the node's documents into the SAME two sets as the
    oracle's -- equal gains in exact arithmetic by construction -- whatever the node's size (the
    walker verifies the sets; sizes are reported, not bounded: round 3's sweeps met 1376
    documents); (iii) the runs cut short stay below the ceiling."""
    from fuzz_parity import sweep
    n = 200
    res = sweep(n, seed, verbose=False)
    assert len(res) == n
    _report_oracle_reruns(res)
    kinds = {}
    for r in res:
        kinds[r["status"]] = kinds.get(r["status"], 0) + 1
    allowed = {"ok", "gain_tie", "gain_tie_fp", "zero_deviance", "heap_tie", "score_tie"}
    assert set(kinds) <= allowed, kinds
    cut = n - kinds.get("ok", 0)
    sizes = [s for r in res for s in r["tie_sizes"]]
    print(f"fuzz seed {seed}: {kinds}, {sum(r['ties'] for r in res)} equal-partition ties (largest node "
          f"{max(sizes or [0])} documents), {sum(r['flips'] for r in res)} runs with a rank flip")
    assert cut <= DIVERGENCE_CEILING * n, (cut, kinds, [r["desc"] for r in res if r["status"] != "ok"])
    for r in res:   # a priced gain tie is exact (0.0) or below the fixed-point resolution
        assert r["status"] == "ok" or ("rest" in r and all(k != "unclassified" for _, k in r["rest"])), \
            (r["desc"], r.get("rest"))   # (the trees behind a cut: verified one by one from the device's own scores)
        if r["status"] in ("gain_tie", "gain_tie_fp") and r["gain_rel"] is not None:
            assert r["gain_rel"] <= 1e-9 and (r["status"] == "gain_tie_fp") == (r["gain_rel"] > 0.0), r
        _assert_fp_tie_is_the_references_rounding(r)


def test_scoring_fuzz_sweep_seed0():
    """240 random ensembles / oblivious ensembles through every scoring kernel (4-byte and 8-byte
    records, u8 and u16 bins, the general fallback, both oblivious scorers): bit-exact against
    the reference's walks restated in numpy (tests/tools/fuzz_scoring.py)."""
    from fuzz_scoring import sweep
    res = sweep(240, 0, verbose=False)
    assert len(res) == 240
    assert all(r["ok"] for r in res), [r["desc"] for r in res if not r["ok"]]
