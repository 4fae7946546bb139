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
