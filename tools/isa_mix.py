#!/usr/bin/env python3
"""ISA audit helper: compiles one .hip translation unit for gfx950 with --save-temps and prints, per kernel, its resources (VGPRs, scratch,
LDS, occupancy) and the instruction mix of every loop (label of the back edge, instructions, VALU, LDS, global/flat memory, waits, s_nop) —
what one looks at for uniform flat_loads, s_waitcnt vmcnt(0) inside column loops, spills and instructions per unit of work.
Needs only hipcc (cross-compiles without a GPU).   Usage: python3 tools/isa_mix.py stwo-brainfuck_amd/csrc/fft.hip [kernel-name-substring ...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    src = os.path.abspath(sys.argv[1])
    want = sys.argv[2:]
    with tempfile.TemporaryDirectory() as d:
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "stwo-brainfuck_amd", "csrc"), "--save-temps", "-c", src, "-o", "x.o"],
                       cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
        L = open(os.path.join(d, asm)).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(L) if re.match(r"^_Z\w+:", l)]
    names = demangle([n for _, n in starts])
    for (a, name), (b, _) in zip(starts, starts[1:] + [(len(L), "")]):
        pretty = names.get(name, name).split("(")[0].replace("void ", "").replace("bf::", "")
        if want and not any(w in pretty for w in want):
            continue
        end = next((i for i in range(a, b) if ".amdhsa_kernel" in L[i]), b)
        res = {}
        for i in range(a, b):
            m = re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|codeLenInByte|LDSByteSize)[:=]? *=? *(\d+)", L[i])
            if m:
                res[m.group(1)] = int(m.group(2))
        if "NumVgprs" not in res:
            continue
        print(f"== {pretty}: VGPRs {res.get('NumVgprs')}, scratch {res.get('ScratchSize')} B, LDS {res.get('LDSByteSize')} B, waves/SIMD {res.get('Occupancy')}, code {res.get('codeLenInByte')} B")
        labels = {L[i].split(":")[0]: i for i in range(a, end) if re.match(r"^\.LBB\d+_\d+:", L[i])}

        def mix(lo, hi):
            ins = [l.strip().split()[0] for l in L[lo:hi] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
            c = collections.Counter(ins)
            g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
            return len(ins), g("v_"), g("ds_"), g("global_"), g("flat_"), g("scratch_"), c.get("s_waitcnt", 0), c.get("s_nop", 0), c
        n, valu, ds, glob, flat, scr, waits, nops, c = mix(a, end)
        print(f"   whole kernel: {n} instructions, {valu} VALU, {ds} LDS, {glob} global, {flat} flat, {scr} scratch, {waits} s_waitcnt, {nops} s_nop")
        for i in range(a, end):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", L[i])
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                n, valu, ds, glob, flat, scr, waits, nops, c = mix(labels[m.group(1)], i + 1)
                if n < 24:
                    continue
                top = ", ".join(f"{k} {v}" for k, v in c.most_common(8))
                print(f"   loop {m.group(1)}: {n} instructions, {valu} VALU, {ds} LDS, {glob} global, {flat} flat, {scr} scratch, {waits} s_waitcnt, {nops} s_nop | {top}")


if __name__ == "__main__":
    main()
