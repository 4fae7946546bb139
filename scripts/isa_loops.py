#!/usr/bin/env python3
"""Static loop structure of one gfx950 kernel of the built library: every backward branch with the
instruction mix of the span it closes (VALU / SALU / LDS / VMEM / waits / branches), from the
disassembly of the code object (no GPU needed).  Dynamic counts come from the PMC passes
(scripts/pmc_summary.py); this says where in the code they can come from.

    python scripts/isa_loops.py k_lambda 'k_lambda<false, 1, true>' [--min 8]

First argument: the source's stem (k_lambda, k_tree, ...); second: the kernel as
scripts/kernel_resources.py prints it."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as K  # noqa: E402

KINDS = (("wait", ("s_waitcnt",)), ("br", ("s_cbranch", "s_branch", "s_endpgm", "s_setpc")), ("valu", ("v_",)),
         ("salu", ("s_",)), ("lds", ("ds_",)), ("vmem", ("global_", "buffer_", "flat_", "scratch_")))


def kind(op):
    for k, pre in KINDS:
        if op.startswith(pre):
            return k
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel")
    ap.add_argument("--lib", default="libqr_hip")
    ap.add_argument("--min", type=int, default=8, help="shortest span listed")
    a = ap.parse_args()
    obj = os.path.join(K.ROOT, "quickrank_amd", "lib", "obj", f"{a.lib}.{a.source}.o")
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "f.bin"), os.path.join(tmp, "k.co")
        subprocess.run([f"{K.LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([f"{K.LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        f"--targets={K.TARGET}", f"--output={co}"], check=True, capture_output=True)
        dis = subprocess.run([f"{K.LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
    heads = [(m.start(), m.group(1)) for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:$", dis, re.M)]
    names = K.demangle([h[1] for h in heads])
    pick = [i for i, n in enumerate(names) if K.short(n) == a.kernel]
    if not pick:
        sys.exit("kernels here: " + ", ".join(sorted({K.short(n) for n in names})))
    i = pick[0]
    body = dis[heads[i][0]:heads[i + 1][0] if i + 1 < len(heads) else len(dis)]
    ins, at = [], {}
    for line in body.splitlines():
        m = re.search(r"//\s*([0-9A-F]{12}):", line)
        if m:
            addr = int(m.group(1), 16)
            at[addr] = len(ins)
            ins.append((addr, line.split("//")[0].strip()))
    total = {}
    for _, t in ins:
        total[kind(t.split()[0])] = total.get(kind(t.split()[0]), 0) + 1
    print(f"{a.kernel}: {len(ins)} instructions, {total}")
    spans = []
    for k, (addr, t) in enumerate(ins):
        m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\d+)", t)
        if m:
            off = int(m.group(2))
            off -= 65536 if off >= 32768 else 0
            tgt = addr + 4 + 4 * off
            if tgt <= addr and tgt in at:
                spans.append((at[tgt], k, m.group(1)))
    for lo, hi, br in sorted(spans):
        if hi - lo + 1 < a.min:
            continue
        mix = {}
        for _, t in ins[lo:hi + 1]:
            mix[kind(t.split()[0])] = mix.get(kind(t.split()[0]), 0) + 1
        print(f"  [{lo:5d} .. {hi:5d}] {hi - lo + 1:5d} instructions, closed by {br:18s} {mix}")


if __name__ == "__main__":
    main()
