#!/usr/bin/env python
"""Builds the code objects of the s_nop experiment (no GPU needed): tools/snop/kernels.hip -> device assembly for 2 and 3 waves
per SIMD, then four variants of each as .hsaco files next to this script:
  as_compiled      what hipcc emits (the compiler's s_nop padding counted per kernel)
  nops_removed     every `s_nop` between two v_mad_u64_u32 deleted -- is the padding needed at all (results) and what does it cost (time)?
  nops_added       an `s_nop 0` after every second v_mad_u64_u32 of the three-chain kernel -- the two-chain kernel's padding density on
                   an instruction stream that does not need it: the cost of the instruction itself
Prints the static counts; tools/snop/run (built here too) loads and times them on the GPU box."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(os.path.dirname(HERE)), "go-snark-study_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_bodies(asm):
    """name -> (start line, end line) of each kernel's instructions"""
    lines = asm.split("\n")
    spans, cur = {}, None
    for i, l in enumerate(lines):
        m = re.match(r"^(k_\w+):", l)
        if m:
            cur = m.group(1)
            spans[cur] = [i, None]
        if cur and l.strip().startswith("s_endpgm"):
            spans[cur][1] = i
            cur = None
    return lines, spans


def assemble(text, out):
    s = out + ".s"
    open(s, "w").write(text)
    subprocess.check_call([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", out + ".o"])
    subprocess.check_call([os.path.join(LLVM, "ld.lld"), "-shared", out + ".o", "-o", out + ".hsaco"])
    os.remove(out + ".o")


def main():
    for waves in (2, 3):
        s_path = os.path.join(HERE, "kernels_w%d.s" % waves)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-DWAVES=%d" % waves, "--cuda-device-only", "-S",
                               os.path.join(HERE, "kernels.hip"), "-o", s_path])
        asm = open(s_path).read()
        lines, spans = kernel_bodies(asm)
        is_mad = lambda l: "v_mad_u64_u32" in l            # noqa: E731
        is_nop = lambda l: re.match(r"\s*s_nop\b", l) is not None     # noqa: E731
        real = lambda l: l.strip() and not l.strip().startswith((";", ".", "//")) and not l.strip().endswith(":")   # noqa: E731
        for name, (a, b) in spans.items():
            body = lines[a:b]
            print("waves %d %-16s instructions %5d  v_mad_u64_u32 %5d  s_nop %4d" % (waves, name, sum(1 for l in body if real(l)),
                  sum(1 for l in body if is_mad(l)), sum(1 for l in body if is_nop(l))))
        assemble(asm, os.path.join(HERE, "w%d_as_compiled" % waves))
        # nops_removed: drop an s_nop whose neighbours (skipping asm markers) are multiply-adds or the shifts / masks of the column end
        out = list(lines)
        (a, b) = spans["k_two_chains"]
        removed = 0
        for i in range(a, b):
            if is_nop(out[i]):
                out[i] = "\t; s_nop removed"
                removed += 1
        print("waves %d: removed %d s_nop from k_two_chains" % (waves, removed))
        assemble("\n".join(out), os.path.join(HERE, "w%d_nops_removed" % waves))
        # nops_added: after every second multiply-add of the three-chain kernel
        out = list(lines)
        (a, b) = spans["k_three_chains"]
        res, k, added = [], 0, 0
        for i, l in enumerate(out):
            res.append(l)
            if a <= i < b and is_mad(l):
                k += 1
                if k % 2 == 0:
                    res.append("\ts_nop 0")
                    added += 1
        print("waves %d: added %d s_nop to k_three_chains" % (waves, added))
        assemble("\n".join(res), os.path.join(HERE, "w%d_nops_added" % waves))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(HERE, "run.hip"), "-o", os.path.join(HERE, "run")])


if __name__ == "__main__":
    main()
