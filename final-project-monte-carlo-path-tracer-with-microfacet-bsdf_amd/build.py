"""Builds csrc/ into libmcpt_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmcpt_hip.so")
SOURCES = ["mcpt_scene.cpp", "mcpt_kernels.hip", "mcpt_api.hip"]
HEADERS = ["mcpt_internal.h", "mcpt_device.h", "mcpt_kernels.h", os.path.join("..", "..", "include", "mcpt.h")]
# -ffp-contract=off: the arithmetic contract of csrc/mcpt_device.h (no FMA contraction, so the same seeds
# give the same paths as the CPU restatement).  f32 divide/sqrt stay correctly rounded (hipcc default).
# -fno-slp-vectorize: the SLP vectoriser pairs scalar f32 adds/muls into v_pk_*_f32, which issue slower than the two
# scalar ops on gfx950 (measured: +1.7 % frame throughput without it); the values are identical either way.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    for s in SOURCES:
        o = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        cmd = [hipcc()] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_host(verbose=False):
    """The C++ host mirror of the reference's executable (host/RayTracing, host/RayTracingDemo)."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host")], stdout=None if verbose else subprocess.DEVNULL)
    return os.path.join(HERE, "host", "RayTracing")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
