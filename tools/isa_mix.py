#!/usr/bin/env python3
"""Instruction mix of the library's kernels from the compiler's own listing - no GPU needed.

    python tools/isa_mix.py [csrc file, default kh_decoder.hip] [-DFLAG ...]

Compiles the file for gfx950 with the library's flags (`hipcc -S`, device side only) and prints, per kernel: instructions,
v_readlane / v_writelane (scalar registers spilled into VGPR lanes: every reload is a VALU issue slot plus wait states -
round 4 found a quarter of DecodeKernel's instructions there), scratch loads / stores and bytes (VGPR spills: a reload
inside a loop waits on vmcnt(0), i.e. on the loop's own stores), ds_bpermute (cross-lane traffic on the LDS pipe) and DPP
instructions.  The listing itself stays in /tmp/<file>.s for a look at the loops (grep for the kernel's mangled name).
What to watch when changing the decoder: scratch bytes of DecodeKernel<1,0> (132 at the end of round 4) - every build that
raised it by 8-24 bytes lost 4-12 % on the GPU."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-result",
         "-D__HIP_PLATFORM_AMD__", "-mllvm", "-amdgpu-inline-max-bb=100000", "-mllvm", "-disable-machine-licm", "--cuda-device-only", "-S"]


def main():
    args = sys.argv[1:]
    src = next((a for a in args if not a.startswith("-")), "kh_decoder.hip")
    defs = [a for a in args if a.startswith("-")]
    path = src if os.path.exists(src) else os.path.join(ROOT, "old-kaldi-git_amd", "csrc", src)
    out = os.path.join("/tmp", os.path.basename(path) + ".s")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc] + FLAGS + defs + [path, "-o", out], check=True, stderr=subprocess.DEVNULL)
    kernels, cur = {}, None
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None:
            kernels[cur].append(line)
    print("%-64s %7s %8s %9s %7s %7s %7s %9s %5s" % ("kernel", "instr", "readlane", "writelane", "scr_ld", "scr_st", "scr_B", "bpermute", "dpp"))
    for name, lines in kernels.items():
        c = collections.Counter()
        scratch = None
        is_kernel = False
        for l in lines:
            m = re.match(r"^\s+([a-z_0-9]+)\s", l)
            if m:
                c[m.group(1)] += 1
            m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", l)
            if m:
                scratch, is_kernel = int(m.group(1)), True
        if not is_kernel:
            continue
        short = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        short = re.sub(r"\(anonymous namespace\)::", "", short).split("(")[0][:64]
        print("%-64s %7d %8d %9d %7d %7d %7s %9d %5d" % (
            short, sum(c.values()), c["v_readlane_b32"], c["v_writelane_b32"],
            sum(v for k, v in c.items() if k.startswith("scratch_load")), sum(v for k, v in c.items() if k.startswith("scratch_store")),
            scratch, c["ds_bpermute_b32"], sum(v for k, v in c.items() if k.endswith("_dpp"))))
    print("listing: " + out)


if __name__ == "__main__":
    main()
