"""Build recipe for libkaldi_hip.so (gfx950 only; hipcc cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkaldi_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-inline-max-bb: the AMDGPU inliner stops inlining into a function that has grown past
# 1100 basic blocks.  The persistent decoder kernel is larger; past the cap its phases become
# real calls that take the arena descriptor by reference, which pins that 400-byte struct in
# scratch memory (every pointer fetched with scratch_load: 1.3x slower kernel, and the cliff
# moved whenever any phase grew).  All device functions of this library are meant to be inlined.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__",
         "-mllvm", "-amdgpu-inline-max-bb=100000"]


# Per-file extras.  kh_decoder.hip: the persistent decode kernels are ONE loop over frames around everything, and the machine-level
# loop-invariant code motion hoists every `threadIdx.x * k + constant` LDS address and every `threadIdx.x < constant` mask to
# the kernel's entry - far more values than the 64-register budget holds, so they are spilled there and reloaded from scratch
# at every use (round 5: kernel-resource-usage ScratchSize 132 -> 84 B per lane for the canonical kernel, 304 -> 204 for the
# reference-order one with the pass off; same-box A/B 554 -> 545 ms and 1210 -> 1167 ms, bit-identical results).
EXTRA = {"kh_decoder.hip": ["-mllvm", "-disable-machine-licm"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, jobs=8):
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    hdr_t = max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > hdr_t):
            continue
        cmd = [HIPCC] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        if len(procs) >= jobs:
            _wait(procs)
    _wait(procs)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


def build_host_test(name="host_api_test"):
    """Compile tests/cpp/<name>.cc (plain g++, host only) against the library: host_api_test (the whole C++
    mirror) or cu_matrix_test (the reference's cu-matrix-test.cc cases, template call syntax)."""
    build()
    root = os.path.normpath(os.path.join(HERE, ".."))
    src = os.path.join(root, "tests", "cpp", name + ".cc")
    exe = os.path.join(HERE, "build", name)
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    deps = [src, LIB] + [os.path.join(HERE, "host", h) for h in ("kaldi-hip.h", "kaldi-matrix-lite.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", src, "-o", exe, LIB, "-Wl,-rpath," + HERE,
                               "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _wait(procs):
    while procs:
        src, p = procs.pop(0)
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)


def build_tools():
    """Standalone HIP programs under tools/ (gfx950): tools/pmc_calibrate.hip -> build/pmc_calibrate, the
    known-byte-count kernels behind profiles/r02_pmc_calibration.txt (tools/pmc_calibrate.sh)."""
    root = os.path.join(HERE, "..")
    src = os.path.join(root, "tools", "pmc_calibrate.hip")
    exe = os.path.join(HERE, "build", "pmc_calibrate")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-o", exe, src])
    return exe

