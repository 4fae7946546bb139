"""The oracle in an address space of its own (TEST TOOL; VERDICT r5 item 1's A/B).

`OracleProc().train(...)` is `oracle.train(...)` run by a child interpreter that imports numpy and
the oracle and NOTHING else -- no torch, no HIP runtime, no device library.  Whatever writes into
host memory of the process that drives the GPU (a stray store of the device library's host code, a
DMA into memory the process no longer owns) cannot reach the oracle's lists there, and whatever
goes wrong in the child (its self-checks, `oracle.SelfCheckError`) happened in a process that never
opened the GPU: either outcome says on which side of the comparison the stray writer of rounds 4-5
lives (profiles/r05_abort_hunt.md, profiles/r06_hunt.md).  The child is started BEFORE the parent
touches the GPU when the caller creates the object early; arguments and results cross a pipe as
pickles (numpy arrays by value)."""
import os
import pickle
import struct
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.abspath(os.path.join(_HERE, "..", ".."))

_SERVER = r"""
import os, pickle, struct, sys
sys.path.insert(0, %r)
import oracle
assert 'torch' not in sys.modules and 'quickrank_amd' not in sys.modules
rd, wr = os.fdopen(int(sys.argv[1]), 'rb'), os.fdopen(int(sys.argv[2]), 'wb')
while True:
    head = rd.read(8)
    if len(head) < 8:
        break
    fn, args, kw = pickle.loads(rd.read(struct.unpack('<Q', head)[0]))
    try:
        res = ('ok', getattr(oracle, fn)(*args, **kw))
    except oracle.SelfCheckError as e:
        res = ('selfcheck', str(e))
    except BaseException as e:
        res = ('error', repr(e))
    maps = open('/proc/self/maps').read()
    assert 'libamdhip64' not in maps and 'libqr_hip' not in maps, 'a HIP runtime crept into the oracle process'
    blob = pickle.dumps(res, protocol=4)
    wr.write(struct.pack('<Q', len(blob))); wr.write(blob); wr.flush()
""" % (_ROOT,)


class OracleProc:
    def __init__(self):
        c2p_r, c2p_w = os.pipe()
        p2c_r, p2c_w = os.pipe()
        env = dict(os.environ)
        for k in ("LD_PRELOAD", "ASAN_OPTIONS", "QR_HIP_LIB"):   # (the child is the plain interpreter)
            env.pop(k, None)
        self.p = subprocess.Popen([sys.executable, "-X", "faulthandler", "-c", _SERVER, str(p2c_r), str(c2p_w)],
                                  pass_fds=(p2c_r, c2p_w), env=env)
        os.close(p2c_r)
        os.close(c2p_w)
        self.wr, self.rd = os.fdopen(p2c_w, "wb"), os.fdopen(c2p_r, "rb")

    def call(self, fn, *args, **kw):
        blob = pickle.dumps((fn, args, kw), protocol=4)
        self.wr.write(struct.pack("<Q", len(blob)))
        self.wr.write(blob)
        self.wr.flush()
        head = self.rd.read(8)
        if len(head) < 8:
            raise RuntimeError(f"the oracle process died (exit status {self.p.wait()})")
        kind, val = pickle.loads(self.rd.read(struct.unpack("<Q", head)[0]))
        if kind == "ok":
            return val
        if kind == "selfcheck":
            import oracle
            raise oracle.SelfCheckError("IN THE ORACLE'S OWN PROCESS (no GPU runtime mapped there): " + val)
        raise RuntimeError("oracle process: " + val)

    def train(self, *args, **kw):
        return self.call("train", *args, **kw)

    def close(self):
        try:
            self.wr.close()
            self.p.wait(timeout=30)
        except Exception:
            self.p.kill()

    def __del__(self):
        self.close()
