"""Exit / teardown stress (TEST TOOL, GPU box): VERDICT r3 item 5a.

One process of the round-3 sweeps (seed 18) ended with a core dump AFTER its work; fifteen
repetitions did not reproduce it.  This tool asks the question at scale: it starts `nproc`
short-lived processes (`jobs` at a time, sharing the GPU), each of which creates and destroys
many device contexts -- leaf-wise, oblivious, ragged query sets (the lambda pass's auxiliary
streams), wide bins, an ensemble upload -- through the randomised sweep of fuzz_parity.py, and
then exits in one of three ways:
    clean      contexts closed explicitly, normal interpreter exit
    leak       the last context is left alive in a global: closed by _capi's atexit hook
    sysexit    sys.exit(0) from inside a function that still holds a live context
Every child runs under `python -X faulthandler`; a non-zero return code (a negative one is a
signal) is a failure and its stderr tail is printed.

    python tests/tools/exit_stress.py [nproc=60] [jobs=4] [configs_per_proc=6]
"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")

CHILD = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, {here!r})
import numpy as np
from fuzz_parity import sweep
mode, seed, ncfg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
res = sweep(ncfg, seed, verbose=False)
assert len(res) == ncfg
from datagen import make_dataset
from quickrank_amd.trainer import Mart
x, labels, qoff = make_dataset(nq=40, docs_per_query=30, F=20, seed=seed)
def run():
    m = Mart(algo="LAMBDAMART", ntrees=2, shrinkage=0.1, nthresholds=32, nleaves=6, minls=1, esr=0)
    m.learn(x, labels, qoff)
    m.score_dataset(x)
    return m
if mode == "clean":
    run().ctx.close()
elif mode == "leak":
    KEEP = run()          # a global: collected (if at all) during interpreter shutdown
else:
    def f():
        m = run()
        sys.exit(0)
    f()
"""


def main():
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ncfg = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    code = CHILD.format(root=os.path.abspath(ROOT), here=HERE)
    modes = ("clean", "leak", "sysexit")
    pending = [(i, modes[i % 3], 1000 + i) for i in range(nproc)]
    running, failures, t0 = [], [], time.time()
    env = dict(os.environ)
    env.setdefault("AMD_LOG_LEVEL", "0")
    while pending or running:
        while pending and len(running) < jobs:
            i, mode, seed = pending.pop(0)
            p = subprocess.Popen([sys.executable, "-X", "faulthandler", "-c", code, mode, str(seed), str(ncfg)],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            running.append((i, mode, seed, p))
        for item in list(running):
            i, mode, seed, p = item
            if p.poll() is None:
                continue
            out, err = p.communicate()
            running.remove(item)
            if p.returncode != 0:
                failures.append((i, mode, seed, p.returncode, err.decode(errors="replace")[-1500:]))
        time.sleep(0.05)
    print(f"exit stress: {nproc} processes ({jobs} at a time, {ncfg} sweep configurations + one training + one scoring "
          f"context each; exits: clean / leaked global / sys.exit with a live context), "
          f"{len(failures)} abnormal exits, {time.time() - t0:.0f} s")
    for i, mode, seed, rc, err in failures:
        print(f"--- process {i} ({mode}, seed {seed}): return code {rc}\n{err}")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
