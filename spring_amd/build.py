"""Builds libspring_reorder_hip.so in-tree with hipcc for gfx950 (no GPU needed to build)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libspring_reorder_hip.so")
SOURCES = ["reorder_kernels.hip", "reorder_pipeline.cpp", "reorder_files.cpp", "order_ops.hip", "fastq_kernels.hip",
           "encoder.hip", "fastq_reorder.hip"]
HEADERS = ["reorder_device.h", "reorder_internal.h", "reorder_round_mc.h", "synth_common.h", "call_reorder.h"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "spring_reorder.h"),
                                                            os.path.join(ROOT, "include", "spring_encoder.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, cmds = [], []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
                os.path.getmtime(os.path.join(CSRC, f)) for f in [src] + HEADERS) and os.path.getmtime(obj) > max(
                os.path.getmtime(os.path.join(ROOT, "include", h)) for h in ("spring_reorder.h", "spring_encoder.h")):
            continue  # object newer than its source and every header
        cmds.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
                     "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", os.path.join(CSRC, src), "-o", obj])
    procs = []
    for cmd in cmds:  # the translation units are independent: compile them side by side
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
