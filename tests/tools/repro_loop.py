"""One configuration of the randomised parity sweep, many times in one process and over several
processes (TEST TOOL, GPU box) -- for a mismatch that shows up once in a while:
    python tests/tools/repro_loop.py CONFIG SEED REPEATS PROCESSES [--debug]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
BODY = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, '..')); sys.path.insert(0, os.path.join(%r, '..', '..'))
from fuzz_parity import sweep
only, seed, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bad = 0
for r in range(reps):
    try:
        sweep(only + 1, seed, only=only, verbose=False)
    except AssertionError as e:
        bad += 1
        print('MISMATCH in repeat', r, str(e)[:200], flush=True)
print('repeats', reps, 'mismatches', bad, flush=True)
""" % (HERE, HERE, HERE)
cfg, seed, reps, procs = (int(v) for v in sys.argv[1:5])
env = dict(os.environ)
if "--debug" in sys.argv:
    env["QR_DEBUG"] = "1"
    env["QR_HIP_LIB"] = os.path.join(HERE, "..", "..", "quickrank_amd", "lib", "libqr_debug.so")
tot = 0
for p in range(procs):
    out = subprocess.run([sys.executable, "-c", BODY, str(cfg), str(seed), str(reps)], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True).stdout
    lines = [l for l in out.splitlines() if "MISMATCH" in l or l.startswith("repeats")]
    print(f"process {p}:", " | ".join(lines[-3:]), flush=True)
