"""End-to-end CLI timing: write a synthetic SVMLight file, train with quicklearn, score
with quickscore; prints the tools' own timing lines (reader MB/s, init, training)."""
import ctypes as C, os, subprocess, sys, tempfile, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from bench import synth
from quickrank_amd import build
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
trees = int(sys.argv[2]) if len(sys.argv) > 2 else 100
build.build(); build.build_host()
x, labels, qoff = synth(nq, 100, 136)
H = C.CDLL(build.HOST_LIB)
sz = C.c_size_t
H.qrh_svml_write.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, sz, sz]
d = tempfile.mkdtemp()
p = os.path.join(d, "train.svml")
t = time.time()
H.qrh_svml_write(p.encode(), x.ctypes.data, labels.ctypes.data, qoff.ctypes.data, len(qoff) - 1, x.shape[1])
print(f"wrote {os.path.getsize(p) / 1e6:.0f} MB ({len(labels)} docs x {x.shape[1]} features) in {time.time() - t:.1f} s")
t = time.time()
r = subprocess.run([os.path.join(ROOT, "quickrank_amd", "bin", "quicklearn"), "--algo", "LAMBDAMART", "--train", p,
                    "--num-trees", str(trees), "--num-leaves", "10", "--num-thresholds", "255", "--shrinkage", "0.1",
                    "--model-out", os.path.join(d, "m.xml")], capture_output=True, text=True)
wall = time.time() - t
keep = [l for l in r.stdout.splitlines() if any(k in l for k in ("Reading time", "Initialization", "Training Time", "on training data"))]
print("\n".join(keep))
print(f"quicklearn wall {wall:.2f} s, exit {r.returncode}")
t = time.time()
r = subprocess.run([os.path.join(ROOT, "quickrank_amd", "bin", "quickscore"), "-d", p, "-m", os.path.join(d, "m.xml"),
                    "-s", os.path.join(d, "s.txt")], capture_output=True, text=True)
print("\n".join(l for l in r.stdout.splitlines() if "time" in l.lower()))
print(f"quickscore wall {time.time() - t:.2f} s, exit {r.returncode}")
