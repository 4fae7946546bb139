#!/usr/bin/env python3
"""Register / LDS / scratch budget of every gfx950 kernel in the built library, read from the
code objects' own metadata (no GPU needed): the static side of "measure, don't guess".

    python scripts/kernel_resources.py [--lib libqr_hip] [--md profiles/rNN_kernel_resources.md]

For every object under quickrank_amd/lib/obj/<lib>.*.o: the .hip_fatbin section is unbundled
(clang-offload-bundler), the gfx950 code object's NT_AMDGPU_METADATA note is parsed, and one row
per kernel is printed: VGPRs (arch + accumulation), SGPRs, spills, static LDS, scratch bytes,
the largest workgroup the kernel is compiled for, and the waves per SIMD the VGPR count allows
(512 registers per lane and SIMD on CDNA3/4, allocated in blocks of 8, at most 8 waves).
Scratch > 0 or VGPR spills > 0 on a hot kernel is the first thing to fix; SGPR spills go to VGPR
lanes (cheap, but they are instructions in the loop they sit in)."""
import argparse
import glob
import os
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return out.stdout.splitlines()


def short(sig):
    """`void k_hist<4, true>(QrFoo, ...)` -> `k_hist<4, true>`"""
    s = sig[5:] if sig.startswith("void ") else sig
    s = s.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(s):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return s[:i]
    return s


def kernels_of(obj, tmp):
    fat, co = os.path.join(tmp, "f.bin"), os.path.join(tmp, "k.co")
    for p in (fat, co):
        if os.path.exists(p):
            os.remove(p)
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat],
                       capture_output=True)
    if r.returncode or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return []           # a host-only object
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    f"--targets={TARGET}", f"--output={co}"], check=True, capture_output=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    a = notes.find("---")
    b = notes.find("...", a)
    if a < 0:
        return []
    meta = yaml.safe_load(notes[a + 3:b if b > 0 else None])
    return meta.get("amdhsa.kernels", [])


def waves_per_simd(vgpr, agpr):
    total = vgpr + agpr          # unified register file on gfx90a+: arch + acc VGPRs share the 512
    alloc = max(8, (total + 7) // 8 * 8)
    return min(8, 512 // alloc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="libqr_hip")
    ap.add_argument("--md", default=None)
    a = ap.parse_args()
    objs = sorted(glob.glob(os.path.join(ROOT, "quickrank_amd", "lib", "obj", a.lib + ".*.o")))
    if not objs:
        sys.exit("no objects: run `python -c 'import __graft_entry__ as g; g.build()'` first")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            ks = kernels_of(o, tmp)
            if not ks:
                continue
            names = demangle([k[".name"] for k in ks])
            for k, nm in zip(ks, names):
                rows.append(dict(src=os.path.basename(o).split(".")[1], kernel=short(nm),
                                 vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0),
                                 sgpr=k.get(".sgpr_count", 0), vspill=k.get(".vgpr_spill_count", 0),
                                 sspill=k.get(".sgpr_spill_count", 0), lds=k.get(".group_segment_fixed_size", 0),
                                 scratch=k.get(".private_segment_fixed_size", 0),
                                 wg=k.get(".max_flat_workgroup_size", 0)))
    rows.sort(key=lambda r: (r["src"], r["kernel"]))
    # rocPRIM / hipCUB instantiations (sorts, selects: set-up work, not the path) are counted, not listed
    lib_rows = [r for r in rows if "rocprim::" in r["kernel"] or "hipcub::" in r["kernel"]]
    rows = [r for r in rows if r not in lib_rows]
    lines = ["| source | kernel | VGPR | AGPR | SGPR | VGPR spills | SGPR spills | static LDS (B) | scratch (B) | max workgroup | waves/SIMD by registers |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| {r['src']} | `{r['kernel']}` | {r['vgpr']} | {r['agpr']} | {r['sgpr']} | {r['vspill']} | "
                     f"{r['sspill']} | {r['lds']} | {r['scratch']} | {r['wg']} | {waves_per_simd(r['vgpr'], r['agpr'])} |")
    bad = [r for r in rows if r["scratch"] or r["vspill"]]
    lib_bad = [r for r in lib_rows if r["scratch"] or r["vspill"]]
    tail = [f"", f"(not listed: {len(lib_rows)} rocPRIM / hipCUB instantiations behind the library sorts and selects of the "
            f"set-up steps, {len(lib_bad)} of them with scratch)", "",
            f"{len(rows)} kernels; {len(bad)} with scratch or VGPR spills"
            + (": " + ", ".join(f"`{r['kernel']}` ({r['scratch']} B)" for r in bad) if bad else "") + ".",
            f"{sum(1 for r in rows if r['sspill'])} with SGPR spills (to VGPR lanes): "
            + ", ".join(f"`{r['kernel']}` ({r['sspill']})" for r in rows if r["sspill"]) + "."]
    text = "\n".join(lines + tail) + "\n"
    if a.md:
        with open(os.path.join(ROOT, a.md) if not os.path.isabs(a.md) else a.md, "w") as f:
            f.write(f"# Kernel resources of `{a.lib}.so` (gfx950 code-object metadata; `scripts/kernel_resources.py`)\n\n")
            f.write("Dynamic LDS (the lambda kernels' per-query arrays, the histogram kernels' bins) is added at launch "
                    "and is not in the static column; DESIGN.md section 3 gives the launch sizes.\n\n")
            f.write(text)
    try:
        print(text)
    except BrokenPipeError:
        pass


if __name__ == "__main__":
    main()
