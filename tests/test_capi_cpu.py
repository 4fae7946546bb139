"""CPU-side checks of the drop-in boundary: the library builds for gfx950,
loads, exports every symbol include/qr_hip.h declares, and refuses to run
without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from quickrank_amd import build, _capi
    build.build()
    return _capi


def test_header_symbols_are_exported(capi):
    hdr = open(os.path.join(ROOT, "include", "qr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(qr_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/qr_hip.h but not exported"
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)


def test_struct_layouts_match_header(capi):
    # qr_node_t: 2xi32, f32, 2xi32, (pad), f64, f64, u64 = 48 B; qr_split_t = 32 B
    assert capi.NODE_DTYPE.itemsize == 48
    assert capi.NODE_DTYPE.fields["value"][1] == 24
    assert capi.SPLIT_DTYPE.itemsize == 32
    assert capi.SPLIT_DTYPE.fields["lcount"][1] == 16


def test_no_cpu_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    rc = capi.lib().qr_ctx_create(0, C.byref(h))
    assert rc == 1 and not h.value  # QR_ERR_NO_DEVICE
    with pytest.raises(capi.QrError):
        capi.Context(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under quickrank_amd/ may touch it."""
    for dp, _, files in os.walk(os.path.join(ROOT, "quickrank_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("quickrank_amd has no CPU fallback", ""), \
                    f"{f} mentions the oracle"


def test_oracle_users_are_only_the_allowed_ones():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch
    oracle/: scripts/ must not, bench.py imports it inside cpu_baseline only, and
    __graft_entry__ outside build() (which compiles the checker) only in smoke()."""
    import ast
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    for dirpath, _, files in os.walk(os.path.join(root, "scripts")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, re.M), f

    def importers(path):
        tree = ast.parse(open(path).read())
        found = set()
        for fn in ast.walk(tree):
            if isinstance(fn, (ast.FunctionDef, ast.Module)):
                for node in (fn.body if isinstance(fn, ast.Module) else ast.walk(fn)):
                    if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                        found.add(getattr(fn, "name", "<module>"))
                    if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                        found.add(getattr(fn, "name", "<module>"))
        return found
    # the cpu_baseline leg: the training baseline and the scoring baseline of the same object
    assert importers(os.path.join(root, "bench.py")) <= {"cpu_baseline", "cpu_scoring_baseline", "cpu_baseline_wide"}
    assert importers(os.path.join(root, "__graft_entry__.py")) <= {"build", "smoke"}
