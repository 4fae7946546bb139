"""Host C++ layer (quickrank_amd/host): SVMLight reader/writer against the
reference's own Svml (oracle/_ref, bit for bit) and the XML model format."""
import ctypes as C
import os

import numpy as np
import pytest

from datagen import make_dataset


@pytest.fixture(scope="module")
def host():
    from quickrank_amd import build
    build.build()
    build.build_host()
    L = C.CDLL(build.HOST_LIB)
    sz = C.c_size_t
    L.qrh_svml_read.argtypes = [C.c_char_p, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.c_void_p,
                                C.c_void_p, C.c_void_p]
    L.qrh_svml_write.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, sz, sz]
    L.qrh_model_roundtrip.argtypes = [C.c_char_p, C.c_char_p]
    L.qrh_model_write.argtypes = [C.c_char_p, C.c_int, sz, C.c_double, sz, sz, sz, sz, sz, C.c_void_p,
                                  sz, sz]
    L.qrh_model_read.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(sz), C.POINTER(sz), sz, sz]
    return L


def _read(fn, path):
    sz = C.c_size_t
    N, F, Q = sz(), sz(), sz()
    fn(path.encode(), C.byref(N), C.byref(F), C.byref(Q), None, None, None)
    x = np.zeros((N.value, F.value), np.float32)
    lab = np.zeros(N.value, np.float32)
    qoff = np.zeros(Q.value + 1, np.uint64)
    fn(path.encode(), C.byref(N), C.byref(F), C.byref(Q), x.ctypes.data, lab.ctypes.data, qoff.ctypes.data)
    return x, lab, qoff


SVML_TEXT = """# a comment line
2 qid:1 1:0.5 3:1.25 # doc one
0 qid:1 2:-3e-2\t4:7
   1   qid:1    1:1 2:2 3:3 4:4 5:5
# another comment
3 qid:2 5:0.125
0 qid:2
1 qid:7 1:1e10 2:-0
2 qid:2 3:9.5 #trailing description 6:1
"""


@pytest.mark.ref
def test_svml_reader_matches_reference(host, oracle_lib, tmp_path):
    R = oracle_lib.ref()
    if R is None:
        pytest.skip("oracle/_ref not present")
    p = str(tmp_path / "a.svml")
    open(p, "w").write(SVML_TEXT)
    a, b = _read(host.qrh_svml_read, p), _read(R.ref_svml_read, p)
    for u, v in zip(a, b):
        assert u.shape == v.shape and np.array_equal(u.view(np.uint32) if u.dtype == np.float32 else u,
                                                     v.view(np.uint32) if v.dtype == np.float32 else v)
    assert a[0].shape == (7, 5) and a[2].tolist() == [0, 3, 5, 6, 7]   # qid change = new query
    # a generated file, written by the reference's writer and by ours: identical bytes
    x, labels, qoff = make_dataset(nq=9, docs_per_query=7, F=11, seed=4, ragged=True, adversarial=True)
    p1, p2 = str(tmp_path / "ref.svml"), str(tmp_path / "ours.svml")
    R.ref_svml_write(p1.encode(), x, labels, qoff, len(qoff) - 1, x.shape[1])
    host.qrh_svml_write(p2.encode(), x.ctypes.data, labels.ctypes.data, qoff.ctypes.data, len(qoff) - 1,
                        x.shape[1])
    assert open(p1, "rb").read() == open(p2, "rb").read()
    rx, rl, rq = _read(host.qrh_svml_read, p2)
    # svml.cc:176-180 writes std::fixed with 9 decimals: exact only to 1e-9 absolute
    assert np.allclose(rx, x, rtol=0, atol=1e-9) and np.array_equal(rl, labels) and np.array_equal(rq, qoff)


def test_number_conversion_is_strtofs(host):
    """sscanf("%f") (svml.cc:111) rounds the text to the nearest float: strtof.  The reader's
    `parse_float` takes a shorter way where the result is certain (one exactly rounded float or
    double operation, Clinger's fast path, with the double-rounding midpoints sent to strtof) and
    must agree with strtof in value bits AND in where the number ends -- on every 0.dddddd, on
    random floats in the writer's %.9f and in %g / %e, on doubles at and next to the midpoints
    between adjacent floats, on random digit strings with exponents, junk and hex prefixes."""
    sz = C.c_size_t
    host.qrh_parse_float_selftest.argtypes = [C.c_uint64, sz, C.c_char_p, sz]
    host.qrh_parse_float_selftest.restype = sz
    first = C.create_string_buffer(128)
    assert host.qrh_parse_float_selftest(7, 150000, first, len(first)) == 0, first.value
    texts = ["16777217", "16777217.0000001", "16777216.9999999", "33554434", "33554435", "0.1", "-0.0", "+.5", "5.",
             "1e", "1e+", "1.e5", ".e5", "0x1p3", "0x", "inf", "nan", "infinity", "-inf", "1e38", "3.5e38", "1e-45",
             "1.17549435e-38", "1.17549428e-38", "00000000000000000000123.4500", "9007199254740993", "9007199254740992e3",
             "4.35", "0.000001", "123456789012345678", "1234567890123456789", "12345678901234567890", "1.5abc", "7#x",
             "8388608.5", "8388609.5", "1.00000005960464477539062", "1.00000005960464477539063", "1.0000000596046447",
             "", ".", "-", "e5", "2.5e-5:", "99999999e-8", "16777215e10", "16777215e11", "1e22", "1e23"]
    buf = b"".join(t.encode() + b"\0" for t in texts)
    host.qrh_parse_float_check.argtypes = [C.c_char_p, sz, C.POINTER(sz)]
    host.qrh_parse_float_check.restype = sz
    bad = sz(0)
    assert host.qrh_parse_float_check(buf, len(texts), C.byref(bad)) == 0, texts[bad.value]


def test_scores_file_is_the_streams_output(host, tmp_path):
    """driver.cc:376-383 / quickscore.cc:122-130 stream one score per line with
    setprecision(max_digits10): "%.17g".  The host formats pieces on all threads and writes once;
    the bytes are those of the stream (checked against C formatting, specials included)."""
    rng = np.random.default_rng(0)
    s = np.concatenate([rng.standard_normal(200_000) * rng.choice([1e-300, 1e-5, 1.0, 1e5, 1e300], 200_000),
                        [0.0, -0.0, 1.0, -1.5, 1e22, 1e23, np.inf, -np.inf, np.nan, 5e-324, 1.7976931348623157e308]])
    p = str(tmp_path / "scores.txt")
    host.qrh_write_scores.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    assert host.qrh_write_scores(p.encode(), s.ctypes.data, len(s)) == 0
    assert open(p).read() == "".join("%.17g\n" % v for v in s)
    assert host.qrh_write_scores(p.encode(), s.ctypes.data, 0) == 0 and open(p).read() == ""
    assert host.qrh_write_scores(str(tmp_path / "no" / "dir.txt").encode(), s.ctypes.data, 3) != 0


def test_svml_reader_standalone(host, tmp_path):
    p = str(tmp_path / "a.svml")
    open(p, "w").write(SVML_TEXT)
    x, lab, qoff = _read(host.qrh_svml_read, p)
    assert x.shape == (7, 5) and lab.tolist() == [2, 0, 1, 3, 0, 1, 2]
    assert x[0].tolist() == [0.5, 0, 1.25, 0, 0] and x[6].tolist() == [0, 0, 9.5, 0, 0]
    # the one-pass handle form (scripts/train_multi_gpu.py): the same arrays
    sz = C.c_size_t
    host.qrh_svml_open.argtypes = [C.c_char_p, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    host.qrh_svml_open.restype = C.c_void_p
    host.qrh_svml_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    host.qrh_svml_close.argtypes = [C.c_void_p]
    N, F, Q = sz(), sz(), sz()
    h = host.qrh_svml_open(p.encode(), C.byref(N), C.byref(F), C.byref(Q))
    x2, l2, q2 = np.empty((N.value, F.value), np.float32), np.empty(N.value, np.float32), np.empty(Q.value + 1, np.uint64)
    host.qrh_svml_copy(h, x2.ctypes.data, l2.ctypes.data, q2.ctypes.data)
    host.qrh_svml_close(h)
    assert np.array_equal(x2, x) and np.array_equal(l2, lab) and np.array_equal(q2, qoff)


def _toy_nodes(capi):
    n = np.zeros((2, 5), capi.NODE_DTYPE)
    n["feature"] = -1
    n["left"] = n["right"] = -1
    n[0, 0] = (2, 7, np.float32(0.1), 1, 2, 0.0, 0.0, 10)
    n[0, 1]["value"] = -1.5
    n[0, 2] = (0, 3, np.float32(3.4028235e38), 3, 4, 0.0, 0.0, 5)
    n[0, 3]["value"] = 0.1
    n[0, 4]["value"] = 1.0 / 3.0
    n[1, 0]["value"] = 2.0
    return n


def test_xml_model_format_and_roundtrip(host, tmp_path):
    from quickrank_amd import _capi
    nodes = _toy_nodes(_capi)
    p = str(tmp_path / "m.xml")
    assert host.qrh_model_write(p.encode(), 1, 100, 0.1, 255, 10, 1, 100, 3, nodes.ctypes.data, 2, 5) == 0
    text = open(p).read()
    want = (
        "<ranker>\n\t<info>\n\t\t<type>LAMBDAMART</type>\n\t\t<trees>100</trees>\n\t\t<leaves>10</leaves>\n"
        "\t\t<shrinkage>0.10000000000000001</shrinkage>\n\t\t<leafsupport>1</leafsupport>\n"
        "\t\t<discretization>255</discretization>\n\t\t<estop>100</estop>\n\t\t<subsample>1</subsample>\n"
        "\t\t<max_features>1</max_features>\n\t\t<collapse_leaves_factor>0</collapse_leaves_factor>\n"
        "\t</info>\n\t<ensemble>\n\t\t<tree id=\"1\" weight=\"0.10000000000000001\">\n\t\t\t<split>\n"
        "\t\t\t\t<feature>3</feature>\n\t\t\t\t<threshold>0.100000001</threshold>\n"
        "\t\t\t\t<split pos=\"left\">\n\t\t\t\t\t<output>-1.5</output>\n\t\t\t\t</split>\n"
        "\t\t\t\t<split pos=\"right\">\n\t\t\t\t\t<feature>1</feature>\n"
        "\t\t\t\t\t<threshold>3.40282347e+38</threshold>\n\t\t\t\t\t<split pos=\"left\">\n"
        "\t\t\t\t\t\t<output>0.10000000000000001</output>\n\t\t\t\t\t</split>\n"
        "\t\t\t\t\t<split pos=\"right\">\n\t\t\t\t\t\t<output>0.33333333333333331</output>\n"
        "\t\t\t\t\t</split>\n\t\t\t\t</split>\n\t\t\t</split>\n\t\t</tree>\n"
        "\t\t<tree id=\"2\" weight=\"0.10000000000000001\">\n\t\t\t<split>\n\t\t\t\t<output>2</output>\n"
        "\t\t\t</split>\n\t\t</tree>\n\t</ensemble>\n</ranker>\n")
    assert text == want
    # load -> save is the identity on the text
    p2 = str(tmp_path / "m2.xml")
    assert host.qrh_model_roundtrip(p.encode(), p2.encode()) == 0
    assert open(p2).read() == text
    # and on the numbers (thresholds to 9, outputs to 17 significant digits: rtnode.cc:58-70)
    sz = C.c_size_t
    nt, mn = sz(), sz()
    host.qrh_model_read(p.encode(), None, None, C.byref(nt), C.byref(mn), 0, 0)
    assert (nt.value, mn.value) == (2, 5)
    back = np.zeros((2, 5), _capi.NODE_DTYPE)
    w = np.zeros(2)
    assert host.qrh_model_read(p.encode(), back.ctypes.data, w.ctypes.data, C.byref(nt), C.byref(mn), 10, 2) == 0
    for k in ("feature", "left", "right"):
        assert np.array_equal(back[k], nodes[k]), k
    assert np.array_equal(back["threshold"].view(np.uint32), nodes["threshold"].view(np.uint32))
    assert np.array_equal(back["value"].view(np.uint64), nodes["value"].view(np.uint64))
    assert w.tolist() == [0.1, 0.1]
    # a model with an XML declaration, comments and pretty spaces (other writers) still parses
    p3 = str(tmp_path / "m3.xml")
    open(p3, "w").write('<?xml version="1.0"?>\n<!-- c -->\n' + text.replace("\t", "  "))
    assert host.qrh_model_roundtrip(p3.encode(), p2.encode()) == 0 and open(p2).read() == text


def test_xml_large_model_parallel_paths(host, tmp_path):
    """A model of a megabyte or more is parsed, built, written and released on all host threads
    (xml.cc: the run of <tree> elements is cut at the element starts, every piece checked to be
    whole <tree> elements ending where the next begins).  The records that come back are the
    records that went in; load + save is the identity on the bytes; and documents the cut cannot
    be trusted on -- a "<tree" inside a comment, a comment between trees, pretty spaces, a
    declaration -- load to the same records through the checks or the serial fallback."""
    import sys
    from quickrank_amd import _capi
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
    from score_bench import make_leafwise_model
    T, leaves = 400, 32
    nodes = np.ascontiguousarray(make_leafwise_model(T, leaves, 50, np.random.default_rng(3))[0])
    p, p2 = str(tmp_path / "big.xml"), str(tmp_path / "big2.xml")
    assert host.qrh_model_write(p.encode(), 1, T, 0.1, 255, leaves, 1, 0, 0, nodes.ctypes.data, T, nodes.shape[1]) == 0
    text = open(p).read()
    assert len(text) > (1 << 20)
    assert host.qrh_model_roundtrip(p.encode(), p2.encode()) == 0 and open(p2).read() == text
    sz = C.c_size_t

    def records(path):
        nt, mn = sz(), sz()
        back = np.zeros((T, nodes.shape[1]), _capi.NODE_DTYPE)
        w = np.zeros(T)
        assert host.qrh_model_read(path.encode(), back.ctypes.data, w.ctypes.data, C.byref(nt), C.byref(mn),
                                   back.size, T) == 0
        assert (nt.value, mn.value) == (T, nodes.shape[1])
        return back

    def same(back):
        # (flatten numbers the nodes in pre-order; the generator in creation order: compare what
        # a walk sees -- every document-independent path: feature, threshold bits, leaf value bits)
        def walk(n, i, out):
            out.append((int(n[i]["feature"]), n[i]["threshold"].tobytes(), n[i]["value"].tobytes() if n[i]["feature"] < 0 else b""))
            if n[i]["feature"] >= 0:
                walk(n, int(n[i]["left"]), out)
                walk(n, int(n[i]["right"]), out)
            return out
        for t in range(0, T, 37):
            assert walk(back[t], 0, []) == walk(nodes[t], 0, []), t

    base = records(p)
    same(base)
    cut = text.index("<tree", text.index("<tree") + 5)     # in front of the second tree
    variants = {
        "comment_with_a_tree_in_it": text[:cut] + '<!-- <tree id="0" weight="1"> -->\n\t\t' + text[cut:],
        "comment_between_trees": text[:cut] + "<!-- nothing -->\n\t\t" + text[cut:],
        "declaration_and_spaces": '<?xml version="1.0"?>\n' + text.replace("\t", "  "),
        "one_line": text.replace("\n", "").replace("\t", ""),
    }
    for name, doc in variants.items():
        pv = str(tmp_path / (name + ".xml"))
        open(pv, "w").write(doc)
        got = records(pv)
        assert got.tobytes() == base.tobytes(), name
    # character data between trees is not a model the cut may accept silently: the serial parser
    # decides (it keeps the trees; the text belongs to <ensemble>, which the loader does not read)
    pv = str(tmp_path / "text_between.xml")
    open(pv, "w").write(text[:cut] + "stray " + text[cut:])
    assert records(pv).tobytes() == base.tobytes()


def test_python_io_mirror(host, tmp_path):
    """quickrank_amd/io.py and Mart.save / Mart.load_model_from_file: the Python host reads and
    writes the reference's formats through the same C++ classes the CLI runs (no GPU here: the
    trainer gets a stand-in for its device context)."""
    from quickrank_amd import _capi, io
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(nq=12, docs_per_query=9, F=7, seed=2, ragged=True)
    p = str(tmp_path / "d.svml")
    io.write_svmlight(p, x, labels, qoff)
    rx, rl, rq = io.read_svmlight(p)
    assert np.allclose(rx, x, rtol=0, atol=1e-9) and np.array_equal(rl, labels) and np.array_equal(rq, qoff)
    s = np.array([0.1, -2.5, 1e-3, 3.0])
    io.write_scores(str(tmp_path / "s.txt"), s)
    assert open(tmp_path / "s.txt").read() == "".join("%.17g\n" % v for v in s)
    # the toy model of test_xml_model_format_and_roundtrip, through the trainer object
    nodes = _toy_nodes(_capi)
    m = Mart(algo="LAMBDAMART", ntrees=100, shrinkage=0.1, nthresholds=255, nleaves=10, minls=1, esr=100, depth=3,
             ctx=object())
    for t in nodes:
        m.ensemble.push(t, 0.1)
    p1, p2 = str(tmp_path / "py.xml"), str(tmp_path / "cc.xml")
    m.save(p1)
    assert host.qrh_model_write(p2.encode(), 1, 100, 0.1, 255, 10, 1, 100, 3, nodes.ctypes.data, 2, 5) == 0
    assert open(p1).read() == open(p2).read()
    back = Mart.load_model_from_file(p1, ctx=object())
    assert (back.algo, back.ntrees, back.nleaves, back.nthresholds, back.minls, back.esr) == ("LAMBDAMART", 100, 10, 255, 1, 100)
    assert back.shrinkage == 0.1 and len(back.ensemble) == 2 and back.ensemble.weights == [0.1, 0.1]
    bn, _ = back.ensemble.arrays()
    for k in ("feature", "left", "right"):
        assert np.array_equal(bn[k], nodes[k]), k
    assert np.array_equal(bn["threshold"].view(np.uint32), nodes["threshold"].view(np.uint32))
    assert np.array_equal(bn["value"].view(np.uint64), nodes["value"].view(np.uint64))
    # per-tree weights survive; an oblivious model keeps its depth; another algorithm's file is None
    m.ensemble.weights = [0.25, 0.5]
    m.save(p1)
    assert Mart.load_model_from_file(p1, ctx=object()).ensemble.weights == [0.25, 0.5]
    o = Mart(algo="OBVMART", ntrees=7, depth=4, nthresholds=16, ctx=object())
    o.ensemble.push(nodes[1], 0.1)
    o.save(p1)
    ob = Mart.load_model_from_file(p1, ctx=object())
    assert (ob.algo, ob.depth, ob.ntrees, len(ob.ensemble)) == ("OBVMART", 4, 7, 1)
    open(p1, "w").write(open(p2).read().replace("LAMBDAMART", "COORDASC"))
    assert Mart.load_model_from_file(p1, ctx=object()) is None


def test_quicklearn_refuses_before_it_needs_a_device(tmp_path):
    """The command line's refusals that come before any device work, on the CPU: the
    reference's exit statuses for malformed SVMLight lines through the real reader (status 1 a
    second token that is not qid:, 2 a line without a label, 3 a negative qid, 4 a malformed
    feature: strutils.cc:68, svml.cc:94-120), an unknown algorithm (driver.cc:58-61), an option of
    a subsystem outside this build's scope, a file that does not exist."""
    import subprocess
    from quickrank_amd import build
    ql = build.build_host()[1]
    for text, status in (("1 quid:3 1:2\n", 1), ("1 qid:1 1:2\n\n2 qid:1 1:3\n", 2), ("1 qid:-4 1:2\n", 3),
                         ("1 qid:3 1:2 x:y\n", 4), ("1 qid:3 0:2\n", 4), ("1 qid:3 2:\n", 4)):
        p = str(tmp_path / "bad.svml")
        open(p, "w").write("# fine\n3 qid:1 1:0.5 2:1\n" * 3 + text)
        r = subprocess.run([ql, "--train", p], capture_output=True, text=True)
        assert r.returncode == status, (text, r.returncode, r.stderr[-200:])
    # ... also when the bad line sits in another thread's chunk of a long file: the FIRST one decides
    p = str(tmp_path / "long.svml")
    good = "".join("%d qid:%d 1:%d.5 2:1 7:3\n" % (i % 5, i // 9 + 1, i) for i in range(60000))
    cut1, cut2 = len(good) // 3, 2 * len(good) // 3
    cut1, cut2 = good.index("\n", cut1) + 1, good.index("\n", cut2) + 1
    open(p, "w").write(good[:cut1] + "1 qid:3 1:2 x:y\n" + good[cut1:cut2] + "1 quid:3 1:2\n" + good[cut2:])
    r = subprocess.run([ql, "--train", p], capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="6"))
    assert r.returncode == 4, r.returncode
    r = subprocess.run([ql, "--algo", "DART", "--train", "x"], capture_output=True, text=True)
    assert r.returncode != 0 and "not set properly" in r.stderr
    r = subprocess.run([ql, "--opt-algo", "CLEAVER"], capture_output=True, text=True)
    assert r.returncode != 0 and "outside this build's scope" in r.stderr
    r = subprocess.run([ql, "--train", str(tmp_path / "nothing.svml")], capture_output=True, text=True)
    assert r.returncode != 0 and "Error while opening file" in r.stderr
    # quickscore: the model comes first (quickscore.cc:84-96)
    qs = build.build_host()[2]
    r = subprocess.run([qs, "-d", p, "-m", str(tmp_path / "nothing.xml")], capture_output=True, text=True)
    assert r.returncode != 0 and "is not parsed correctly" in r.stderr
    other = str(tmp_path / "other.xml")
    open(other, "w").write("<ranker><info><type>COORDASC</type></info><ensemble></ensemble></ranker>")
    r = subprocess.run([qs, "-d", p, "-m", other], capture_output=True, text=True)
    assert r.returncode != 0 and "unsupported model type" in r.stderr
    r = subprocess.run([qs, "-d", p], capture_output=True, text=True)
    assert r.returncode != 0 and "quickscore -d <dataset> -m <model.xml>" in (r.stdout + r.stderr)


def test_command_lines_fail_loudly_without_a_device(tmp_path):
    """No CPU fallback anywhere: with valid input and no GPU in sight, `quicklearn` and
    `quickscore` read their files, then stop with the library's message and a failure status."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is visible here")
    from quickrank_amd import _capi, build
    _, ql, qs = build.build_host()
    p = str(tmp_path / "good.svml")
    open(p, "w").write("".join("%d qid:%d 1:%d.5 2:1 7:3\n" % (i % 5, i // 9 + 1, i) for i in range(200)))
    r = subprocess.run([ql, "--train", p, "--num-trees", "2"], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device visible (this library has no CPU fallback)" in r.stdout + r.stderr
    assert "Dataset size: 200 x 7" in r.stdout            # (the reader had done its part)
    nodes = _toy_nodes(_capi)
    L = C.CDLL(build.HOST_LIB)
    m = str(tmp_path / "m.xml")
    L.qrh_model_write.argtypes = [C.c_char_p, C.c_int, C.c_size_t, C.c_double] + [C.c_size_t] * 5 + [C.c_void_p] + [C.c_size_t] * 2
    assert L.qrh_model_write(m.encode(), 1, 100, 0.1, 255, 10, 1, 100, 3, nodes.ctypes.data, 2, 5) == 0
    r = subprocess.run([qs, "-d", p, "-m", m], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device visible" in r.stdout + r.stderr


def test_oblivious_xml_info_block(host, tmp_path):
    from quickrank_amd import _capi
    nodes = _toy_nodes(_capi)
    p = str(tmp_path / "o.xml")
    host.qrh_model_write(p.encode(), 3, 50, 0.05, 16, 8, 2, 7, 3, nodes.ctypes.data, 1, 5)
    t = open(p).read()
    # obliviouslambdamart.cc:72-83: <depth> after <leaves>, estop carries nthresholds, no subsample block
    assert "<type>OBVLAMBDAMART</type>" in t and "<leaves>8</leaves>\n\t\t<depth>3</depth>" in t
    assert "<estop>16</estop>" in t and "subsample" not in t


_READER_PROBE = r"""
import ctypes as C, sys, json, hashlib
import numpy as np
lib, fn, path = sys.argv[1], sys.argv[2], sys.argv[3]
L = C.CDLL(lib)
f = getattr(L, fn)
sz = C.c_size_t
f.argtypes = [C.c_char_p, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.c_void_p, C.c_void_p, C.c_void_p]
N, F, Q = sz(), sz(), sz()
f(path.encode(), C.byref(N), C.byref(F), C.byref(Q), None, None, None)
x = np.zeros((N.value, F.value), np.float32); lab = np.zeros(N.value, np.float32)
qoff = np.zeros(Q.value + 1, np.uint64)
f(path.encode(), C.byref(N), C.byref(F), C.byref(Q), x.ctypes.data, lab.ctypes.data, qoff.ctypes.data)
h = hashlib.sha1(x.tobytes() + lab.tobytes() + qoff.tobytes()).hexdigest()
print(json.dumps([N.value, F.value, Q.value, h]))
"""

# (text, exit status the reference ends with; 0 = parses)
SVML_EDGE = [
    ("2 qid:1 1:0.5 3:0.25\n0 qid:1 2:1\n", 0),
    ("2 qid:1 1:0.5 3:0.25", 0),                       # no newline at the end
    ("\t 3 qid:7  4:1e-3\r\n1 qid:7 1:1\r\n", 0),      # CRLF, leading blanks
    ("1 qid:1 2:1 2:5 1:0x1p-3 3:inf\n", 0),           # duplicates (last wins), hex float, inf
    ("1 qid:1 1:0.5 #tail 2:1\n", 0),                  # trailing description
    ("1 qid:1 1:0.5junk 2:1\n", 0),                    # junk after a number is ignored
    ("# only\n#comments\n", 0),
    ("1 qid:3\n1 qi\n2\n", 0),                         # no features / truncated qid token / label only
    ("1 qid:1 1:1\n\n2 qid:1 1:2\n", 2),               # blank line: the label is mandatory
    ("1 qid:1 1:1\n   \t \n", 2),
    ("1 2:3 4:5\n", 1),                                # no qid
    ("1 qid:-5 1:1\n", 3),
    ("1 qid:1 abc\n", 4),
    ("1 qid:1 1:\n", 4),
    ("1 qid:1 1: 0.5\n", 4),                           # value in the next token
    ("1 qid:1 1:0.5#tail 2:1\n", 4),                   # '#' inside a token is swallowed
    ("1 qid:1 3\n", 4),
    ("0 qid:1 1:1\n1 qid:1 :5\n1 qid:9 x\n", 4),       # the FIRST malformed line decides
]


@pytest.mark.ref
@pytest.mark.parametrize("case", range(len(SVML_EDGE)))
def test_svml_edge_cases_match_reference(host, oracle_lib, tmp_path, case):
    import subprocess
    import sys
    from quickrank_amd import build
    if oracle_lib.ref() is None:
        pytest.skip("oracle/_ref not present")
    ref_lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "libqr_ref.so")
    text, status = SVML_EDGE[case]
    p = str(tmp_path / "e.svml")
    open(p, "wb").write(text.encode())
    outs = []
    for lib, fn in ((build.HOST_LIB, "qrh_svml_read"), (ref_lib, "ref_svml_read")):
        r = subprocess.run([sys.executable, "-c", _READER_PROBE, lib, fn, p], capture_output=True, text=True)
        outs.append((r.returncode, r.stdout.strip()))
    assert outs[0] == outs[1], (text, outs)
    assert outs[0][0] == status


@pytest.mark.ref
def test_svml_parallel_reader_large_file(host, oracle_lib, tmp_path):
    """Enough text for one chunk per thread: ragged rows, comments and descriptions
    sprinkled in, lines of very different lengths."""
    R = oracle_lib.ref()
    if R is None:
        pytest.skip("oracle/_ref not present")
    rng = np.random.default_rng(5)
    lines = []
    for i in range(40000):
        if i % 97 == 0:
            lines.append("# comment %d" % i)
        nf = int(rng.integers(0, 40))
        ids = np.sort(rng.choice(np.arange(1, 137), nf, replace=False))
        feats = " ".join("%d:%.7g" % (f, v) for f, v in zip(ids, rng.standard_normal(nf) * 10.0 ** rng.integers(-3, 4)))
        tail = " # doc%d" % i if i % 5 == 0 else ""
        lines.append("%d qid:%d %s%s" % (rng.integers(0, 5), i // 37, feats, tail))
    p = str(tmp_path / "big.svml")
    open(p, "w").write("\n".join(lines) + "\n")
    os.environ["OMP_NUM_THREADS"] = "6"
    a, b = _read(host.qrh_svml_read, p), _read(R.ref_svml_read, p)
    for u, v in zip(a, b):
        assert u.shape == v.shape and np.array_equal(u.view(np.uint32) if u.dtype == np.float32 else u,
                                                     v.view(np.uint32) if v.dtype == np.float32 else v)
    # ... and written back by both writers (ours formats pieces of rows on all threads): same bytes
    x, lab, qoff = a
    p1, p2 = str(tmp_path / "ref_out.svml"), str(tmp_path / "our_out.svml")
    R.ref_svml_write(p1.encode(), x, lab, qoff, len(qoff) - 1, x.shape[1])
    host.qrh_svml_write(p2.encode(), x.ctypes.data, lab.ctypes.data, qoff.ctypes.data, len(qoff) - 1, x.shape[1])
    assert os.path.getsize(p2) > (16 << 20) and open(p1, "rb").read() == open(p2, "rb").read()


# ---------------------------------------------------------------------------
# Code generators (driver.cc:197-224): known answers derived from the reference's
# printing rules, and the emitted C compiled and run against an independent walk.
# The reference's own generators need pugixml (un-vendored): parity unpinned.
def _codegen(host, generator, model, out):
    host.qrh_codegen.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    return host.qrh_codegen(generator.encode(), str(model).encode(), str(out).encode())


def _toy_model(host, tmp_path):
    from quickrank_amd import _capi
    p = tmp_path / "toy.xml"
    host.qrh_model_write(str(p).encode(), 1, 100, 0.1, 255, 10, 1, 100, 3, _toy_nodes(_capi).ctypes.data, 2, 5)
    return p


def test_codegen_condop_known_answer(host, tmp_path):
    out = tmp_path / "r.c"
    assert _codegen(host, "condop", _toy_model(host, tmp_path), out) == 0
    # weight through a float with 3 decimals + "f"; threshold text + "f"; leaves verbatim
    assert open(out).read() == (
        "double ranker(float* v) {\n\treturn 0.0 \n"
        "\t\t + 0.100f * ( v[2] <= 0.100000001f ? -1.5 : ( v[0] <= 3.40282347e+38f ? "
        "0.10000000000000001 : 0.33333333333333331 ) )\n"
        "\t\t + 0.100f * 2;\n}\n")
    # an integer-looking threshold gets ".0" before the suffix
    xml = open(_toy_model(host, tmp_path)).read().replace("<threshold>0.100000001</threshold>",
                                                          "<threshold> 7 </threshold>")
    m2 = tmp_path / "int.xml"
    open(m2, "w").write(xml)
    assert _codegen(host, "condop", m2, out) == 0 and "v[2] <= 7.0f ?" in open(out).read()


def test_codegen_vpred_known_answer(host, tmp_path):
    out = tmp_path / "v.txt"
    assert _codegen(host, "vpred", _toy_model(host, tmp_path), out) == 0
    # breadth first; ids below 2^depth - 1 are inner slots (a leaf there is printed as a
    # node carrying its parent's feature); leaf values are shrinkage * output in %g
    assert open(out).read() == (
        "2\n"
        "2\nroot 0 2 0.100000001\nnode 1 0 2 1 -0.15\nnode 2 0 0 0 3.40282347e+38\n"
        "leaf 3 2 1 0.01\nleaf 4 2 0 0.0333333\nend\n"
        "0\nleaf 0 4294967295 0 0.2\nend\n")
    bad = tmp_path / "bad.xml"
    open(bad, "w").write("<ranker><info>")
    assert _codegen(host, "vpred", bad, out) == 1
    assert _codegen(host, "vpred", "", out) == 1


def _compile_ranker(src, tmp_path, name):
    import subprocess
    so = str(tmp_path / (name + ".so"))
    subprocess.check_call(["gcc", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-x", "c", str(src), "-o", so])
    L = C.CDLL(so)
    L.ranker.restype = C.c_double
    L.ranker.argtypes = [C.c_void_p]
    return L


def _scripts():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
    import score_bench
    return score_bench


def test_codegen_condop_runs_like_the_ensemble(host, tmp_path):
    sb = _scripts()
    rng = np.random.default_rng(5)
    nodes, w = sb.make_model(40, 4, 12, rng)
    # ragged trees: cut some subtrees to leaves
    for t in range(0, 40, 3):
        nodes[t, 2]["feature"] = -1
        nodes[t, 2]["value"] = rng.standard_normal()
    m = tmp_path / "m.xml"
    host.qrh_model_write(str(m).encode(), 1, 40, 0.1, 255, 16, 1, 100, 3, nodes.ctypes.data, 40, nodes.shape[1])
    src = tmp_path / "ranker.c"
    assert _codegen(host, "condop", m, src) == 0
    L = _compile_ranker(src, tmp_path, "condop")
    x = rng.random((500, 12), dtype=np.float32)
    got = np.array([L.ranker(x[i].ctypes.data) for i in range(len(x))])
    wf = np.full(40, np.float64(np.float32(0.1)))            # "0.100f"
    assert np.array_equal(got, sb.numpy_score(nodes, wf, x))


def _oblivious_nodes(T, depths, F, rng, capi):
    """heap-ordered symmetric trees (ot.cc:139-140); depths[t] levels for tree t"""
    D = max(depths)
    nn = (1 << (D + 1)) - 1
    nodes = np.zeros((T, nn), capi.NODE_DTYPE)
    nodes["feature"] = -1
    nodes["left"] = nodes["right"] = -1
    for t in range(T):
        d = depths[t]
        f = rng.integers(0, F, d)
        th = rng.random(d, dtype=np.float32)
        for i in range((1 << d) - 1):
            lvl = int(np.log2(i + 1))
            nodes[t, i]["feature"] = f[lvl]
            nodes[t, i]["threshold"] = th[lvl]
            nodes[t, i]["left"] = 2 * i + 1
            nodes[t, i]["right"] = 2 * i + 2
        nodes["value"][t, (1 << d) - 1:(1 << (d + 1)) - 1] = rng.standard_normal(1 << d)
    return nodes


@pytest.mark.parametrize("T,mixed", [(12, False), (12, True), (60, True)])
def test_codegen_oblivious_runs_like_the_ensemble(host, tmp_path, T, mixed):
    from quickrank_amd import _capi
    sb = _scripts()
    rng = np.random.default_rng(6 + T)
    depths = [int(d) for d in (rng.integers(1, 5, T) if mixed else np.full(T, 4))]
    if mixed:
        depths[0] = 4
    nodes = _oblivious_nodes(T, depths, 9, rng, _capi)
    m = tmp_path / "o.xml"
    host.qrh_model_write(str(m).encode(), 3, T, 0.05, 16, 16, 1, 100, 4, nodes.ctypes.data, T, nodes.shape[1])
    src = tmp_path / "obl.c"
    assert _codegen(host, "oblivious", m, src) == 0
    text = open(src).read()
    assert text.startswith(f"#define N {T} // no. of trees\n#define M 4 // max tree depth\n"
                           "#define F 16 // max number of leaves\n\nconst float tree_weights[N] = { 0.050000001, ")
    # one scoring loop per depth, shallow trees first, populations adding up to N
    import re
    pops = [int(v) for v in re.findall(r"for \(int j = 0; j < (\d+); \+\+j\)", text)]
    assert sum(pops) == T and len(pops) == max(depths)
    assert pops == [depths.count(d) for d in range(1, max(depths) + 1)]
    L = _compile_ranker(src, tmp_path, f"obl{T}{mixed}")
    x = rng.random((400, 9), dtype=np.float32)
    got = np.array([L.ranker(x[i].ctypes.data) for i in range(len(x))])
    wf = np.full(T, np.float64(np.float32(0.05)))
    want = sb.numpy_score(nodes, wf, x)                      # bit = 1 <=> x > threshold <=> right
    if T <= 16 and not mixed:
        assert np.array_equal(got, want)                     # same tree order, same sum
    else:
        assert np.allclose(got, want, rtol=1e-13, atol=1e-13)  # trees regrouped by depth


def test_codegen_through_quicklearn_flags(host, tmp_path):
    """`quicklearn --model-file X --code-file Y [--generator G]` alone generates and
    exits 0 (driver.cc:47-51, 197-224); without --code-file nothing is written."""
    import subprocess
    from quickrank_amd import build
    exe = os.path.join(build.BINDIR, "quicklearn")
    m = _toy_model(host, tmp_path)
    out = tmp_path / "cli.c"
    r = subprocess.run([exe, "--model-file", str(m), "--code-file", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and "applying conditional operators strategy for C code generation to: " in r.stdout
    assert open(out).read().startswith("double ranker(float* v) {")
    out2 = tmp_path / "cli.v"
    r = subprocess.run([exe, "--model-file", str(m), "--code-file", str(out2), "--generator", "vpred"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and open(out2).read().startswith("2\n2\nroot 0 2 ")
    r = subprocess.run([exe, "--model-file", str(m)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == ""


def test_multi_gpu_query_slices_are_balanced_and_never_empty(host):
    """`quicklearn --gpus W --shard docs` (host/mart_multi.cc): the ranks' query slices tile
    [0, Q) in order, differ by at most one query, and no rank is empty while W <= Q
    (ceil(Q / W)-sized slices left rank 3 of 4 without queries at Q = 9: ADVICE r2)."""
    sz = C.c_size_t
    host.qrh_query_slice.argtypes = [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]
    host.qrh_query_slice.restype = None
    for Q in (1, 2, 7, 8, 9, 10, 63, 64, 65, 6000, 10000):
        for W in (1, 2, 3, 4, 7, 8):
            if W > Q:
                continue
            prev, sizes = 0, []
            for r in range(W):
                a, b = sz(), sz()
                host.qrh_query_slice(Q, r, W, C.byref(a), C.byref(b))
                assert a.value == prev and b.value > a.value, (Q, W, r)
                sizes.append(b.value - a.value)
                prev = b.value
            assert prev == Q and max(sizes) - min(sizes) <= 1, (Q, W, sizes)
