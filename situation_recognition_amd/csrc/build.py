#!/usr/bin/env python3
"""Builds libsrhip.so (gfx950 only) next to the package: hipcc, one object per .hip file,
linked into situation_recognition_amd/libsrhip.so.  No torch, no cmake.  Cross-compiles
without a GPU.  Usage: python situation_recognition_amd/csrc/build.py [--force]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libsrhip.so")
OBJ = os.path.join(HERE, "_obj")
SOURCES = ["gemm.hip", "gram.hip", "elementwise.hip", "ggnn.hip", "expand.hip", "fp8.hip", "stem.hip", "c3d.hip", "c3ds.hip", "pair.hip", "comm.hip"]
HEADERS = [os.path.join(HERE, "common.h"), os.path.join(os.path.dirname(PKG), "include", "srhip.h")]
# -pragma-unroll-threshold: the GEMM epilogues are fully unrolled over 32 accumulator fragments; LLVM's default
# threshold (16K) silently downgrades "#pragma unroll" to a partial unroll, which makes the accumulator index dynamic and
# sends the accumulators to scratch memory.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
         "-mllvm", "-pragma-unroll-threshold=400000"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(HERE, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(OUT, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"])
    return OUT


def build_stamps():
    """Diagnostic library with in-kernel cycle stamps in gemm.hip (-DSR_STAMPS, read back with
    sr_debug_stamps under SR_GEMM_DEBUG=4): libsrhip_stamps.so; select it with SR_LIB_PATH."""
    build(verbose=False)
    hipcc = _hipcc()
    stamped = {"gemm.hip": "-DSR_STAMPS"}
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES if s not in stamped]
    jobs = []
    for src, flag in stamped.items():
        obj = os.path.join(OBJ, src.replace(".hip", "_stamps.o"))
        objs.append(obj)
        jobs.append([hipcc] + FLAGS + [flag, "-c", os.path.join(HERE, src), "-o", obj])
    with ThreadPoolExecutor(max_workers=2) as ex:
        list(ex.map(lambda c: subprocess.run(c, check=True), jobs))
    out = os.path.join(PKG, "libsrhip_stamps.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"], check=True)
    return out


if __name__ == "__main__":
    print(build_stamps() if "--stamps" in sys.argv else build(force="--force" in sys.argv))
