"""Builds csrc/ into libmcpt_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmcpt_hip.so")
# The checking build: the same sources with the direct-lighting-skip check and the pure test hooks (MCPT_RING_START,
# MCPT_HOST_DELAY_US) compiled in, and the retry flavour of the traversal stack forced for every tree with only 4 LDS
# entries (so that most rays lose an entry, go through the retrace lists and are traced again with the scratch stack).  Tests load it explicitly; the product library carries none of it.
LIB_CHECK = os.path.join(HERE, "libmcpt_hip_check.so")
CHECK_DEFINES = ["-DMCPT_TEST_HOOKS", "-DMCPT_CHECK_DIRECT_SKIP", "-DMCPT_FORCE_RETRY", "-DMCPT_STK_RETRY=4"]
SOURCES = ["mcpt_scene.cpp", "mcpt_kernels.hip", "mcpt_api.hip", "mcpt_multi.hip", "mcpt_lbvh.hip", "mcpt_cull.hip"]
HEADERS = ["mcpt_internal.h", "mcpt_device.h", "mcpt_fmath.h", "mcpt_kernels.h", "mcpt_lbvh.h", "mcpt_cull.h", os.path.join("..", "..", "include", "mcpt.h")]
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


def source_tag():
    """Identifies the kernel sources a library was built from: sha1 over csrc/ sources, headers and the compiler flags (12 hex digits).
    profiles/traffic.json records it, and bench.py applies that profile only to the build it was measured on."""
    import hashlib
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(lib, defines, suffix, verbose):
    from concurrent.futures import ThreadPoolExecutor

    def one(s):
        o = os.path.join(CSRC, os.path.splitext(s)[0] + suffix + ".o")
        cmd = [hipcc()] + FLAGS + defines + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(one, SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=False):
    if not force and not needs_build(LIB):
        return LIB
    return _compile(LIB, [], "", verbose)


def build_check(force=False, verbose=False):
    """libmcpt_hip_check.so (see LIB_CHECK)."""
    if not force and not needs_build(LIB_CHECK):
        return LIB_CHECK
    return _compile(LIB_CHECK, CHECK_DEFINES, ".check", verbose)


def build_host(verbose=False):
    """The C++ host mirror of the reference's executable (host/RayTracing, host/RayTracingDemo)."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host")], stdout=None if verbose else subprocess.DEVNULL)
    return os.path.join(HERE, "host", "RayTracing")


def build_variant(name, defines, verbose=False):
    """An experimental build for A/B measurements (tools/ab.sh): libmcpt_hip_<name>.so with extra -D flags."""
    return _compile(os.path.join(HERE, "libmcpt_hip_%s.so" % name), list(defines), "." + name, verbose)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=False))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_check(force="--force" in sys.argv, verbose=True))
