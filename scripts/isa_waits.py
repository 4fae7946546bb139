"""Loads that wait for each other (TEST TOOL, runs here: hipcc cross-compiles).  Compiles a .hip
file of csrc/ to gfx950 assembly and prints, per kernel, the order of global loads (L), full
drains `s_waitcnt vmcnt(0)` (|) and barriers (B), and how many loads sit ALONE between two drains:
a loop written `x = a[i]; y = b[x]` per element compiles to L|L|L|L|..., one round trip after the
other, where `all a[i], then all b[x]` compiles to LLLL....|LLLL....| -- the partition spent most of
its time that way until the round-3 scan.
    python scripts/isa_waits.py k_tree.hip [min_single_loads]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(ROOT, "quickrank_amd", "csrc", sys.argv[1] if len(sys.argv) > 1 else "k_tree.hip")
least = int(sys.argv[2]) if len(sys.argv) > 2 else 4
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17",
                       "-ffp-contract=off", "--cuda-device-only", "-S", "-o", out, src])
lines = open(out).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for n, (i, name) in enumerate(starts):
    end = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    seq = []
    for b in lines[i:end]:
        b = b.strip()
        if b.startswith("global_load") or b.startswith("buffer_load"):
            seq.append("L")
        elif b.startswith("s_waitcnt") and "vmcnt(0)" in b:
            seq.append("|")
        elif b.startswith("s_barrier"):
            seq.append("B")
    t = "".join(seq)
    single = len(re.findall(r"\|L(?=\|)", t))
    if single >= least:
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        print(f"{name.split('(')[0]}: {t.count('L')} loads, {single} alone between two drains\n  {t[:300]}")
