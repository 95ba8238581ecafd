#!/usr/bin/env python3
"""Where a kernel's VGPR spills are: scratch loads / stores and instructions per source-line bucket, from a listing with
line tables (no GPU needed).

    python tools/scratch_by_line.py [kernel-name-substring, default DecodeKernelILb1ELb1] [bucket, default 25] [-DFLAG ...]

Every scratch_load in a loop is a memory round trip behind an s_waitcnt vmcnt(0) that also waits for the loop's own stores
(DESIGN: the round-5 scans).  The listing stays in /tmp/kh_dec_g.s."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-result",
         "-D__HIP_PLATFORM_AMD__", "-mllvm", "-amdgpu-inline-max-bb=100000", "-mllvm", "-disable-machine-licm", "-gline-tables-only", "--cuda-device-only", "-S"]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-D")]
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    name = args[0] if args else "DecodeKernelILb1ELb1"
    bucket = int(args[1]) if len(args) > 1 else 25
    src = os.path.join(ROOT, "old-kaldi-git_amd", "csrc", "kh_decoder.hip")
    out = "/tmp/kh_dec_g.s"
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + defs + [src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if name in l and l.startswith("_Z") and ":" in l.split()[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cur = 0
    ld, st, ins = collections.Counter(), collections.Counter(), collections.Counter()
    for l in lines[start:end]:
        m = re.match(r"\s+\.loc\s+\d+\s+(\d+)", l)
        if m:
            cur = int(m.group(1)) // bucket * bucket
            continue
        m = re.match(r"\s+([a-z_0-9]+)\s", l)
        if m:
            ins[cur] += 1
            if m.group(1).startswith("scratch_load"):
                ld[cur] += 1
            if m.group(1).startswith("scratch_store"):
                st[cur] += 1
    print("source line  scratch_ld  scratch_st  instructions   (total %d loads, %d stores, %d instructions)" % (sum(ld.values()), sum(st.values()), sum(ins.values())))
    for k in sorted(ins):
        if ld[k] + st[k] >= 4:
            print("%9d %10d %10d %12d" % (k, ld[k], st[k], ins[k]))


if __name__ == "__main__":
    main()
