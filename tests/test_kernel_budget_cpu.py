"""The register / scratch budget of the hot kernels, read from the built code objects (no GPU):
a change that pushes one of them into scratch or over an occupancy step shows up here, on the CPU,
before any timing does.  The numbers are those of profiles/r05_kernel_resources.md
(scripts/kernel_resources.py); ceilings, not equalities -- fewer registers are welcome."""
import glob
import os
import sys
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "scripts"))

# kernel -> (max VGPRs, max scratch bytes)
BUDGET = {
    "k_hist_root": (128, 8),           # the roofline's kernel: at its 1024-thread cap, one spill outside the loop
    "k_hist": (128, 0),
    "k_hist_batch": (96, 0),           # 5 waves per SIMD
    "k_hist_level": (128, 0),
    "k_lambda<false, 1, true>": (80, 0),    # queries of <= 128 documents: 6 waves per SIMD
    "k_lambda<false, 1, false>": (80, 0),
    "k_lambda_u": (96, 0),             # 5 waves per SIMD, two eight-wave workgroups per CU by LDS
    "k_partition_batch": (64, 0),
    "k_decide_part<96>": (96, 0),
    "k_leaf_sums_doc<false>": (48, 0),
    "k_score_update_leaf": (24, 0),
    "k_score_p4<16, 256>": (64, 0),
    "k_obl_score_s<8, 6>": (64, 0),
}


@pytest.fixture(scope="module")
def rows():
    from quickrank_amd import build
    build.build()
    import kernel_resources as K
    objs = sorted(glob.glob(os.path.join(HERE, "..", "quickrank_amd", "lib", "obj", "libqr_hip.*.o")))
    if not objs:
        pytest.skip("no object files here (they stay in the build container)")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            ks = K.kernels_of(o, tmp)
            if not ks:
                continue
            for k, nm in zip(ks, K.demangle([k[".name"] for k in ks])):
                out[K.short(nm)] = k
    return out


def test_hot_kernels_stay_in_their_budget(rows):
    for name, (vgpr, scratch) in BUDGET.items():
        assert name in rows, (name, "not in the library any more: update the table")
        k = rows[name]
        assert k.get(".vgpr_count", 0) + k.get(".agpr_count", 0) <= vgpr, (name, k.get(".vgpr_count"))
        assert k.get(".private_segment_fixed_size", 0) <= scratch, (name, k.get(".private_segment_fixed_size"))


def test_no_kernel_of_ours_spills_vector_registers_but_the_known_ones(rows):
    ours = {n: k for n, k in rows.items() if "rocprim::" not in n and "hipcub::" not in n}
    assert len(ours) >= 140
    spilling = {n: (k.get(".vgpr_spill_count", 0), k.get(".private_segment_fixed_size", 0)) for n, k in ours.items()
                if k.get(".vgpr_spill_count", 0) or k.get(".private_segment_fixed_size", 0)}
    assert set(spilling) <= {"k_hist_root", "k_obl_plan"}, spilling
