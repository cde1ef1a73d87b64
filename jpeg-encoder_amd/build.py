"""Builds the native libraries of the package in-tree.

  libjpeg_hip.so   HIP kernels + C ABI (include/jpeg_hip.h), gfx950 only
  libjpeg_host.so  host-side BMP reader and JFIF writer (include/jpeg_host.h)
  jpeg-encoder     C++ command-line front end with the reference's flags

Run as `python jpeg-encoder_amd/build.py` or through __graft_entry__.build().
hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXX = os.environ.get("CXX") or shutil.which("g++") or "g++"

# -ffp-contract=off: the reference's f32 arithmetic is never fused; fused
# multiply-adds are written explicitly (fmaf) where they are wanted.
# -fno-slp-vectorize: packed f32 instructions have no throughput advantage on
# gfx950 (a wave64 v_pk_fma_f32 takes twice the cycles of v_fma_f32) and the
# vectoriser pays for them with register shuffles.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function",
             "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
HIP_SOURCES = ["front.hip", "coded.hip", "staged.hip", "entropy.hip", "api.hip", "static.hip", "chain.hip"]
HIP_HEADERS = ["internal.hpp", "transform.hpp", "tile.hpp", "coding.hpp", "chain.hpp", "tables.h", os.path.join(ROOT, "include", "jpeg_hip.h")]
LIB_HIP = os.path.join(HERE, "libjpeg_hip.so")

HOST_SOURCES = ["host/bmp.cpp", "host/jfif.cpp", "host/stripes.cpp"]
HOST_HEADERS = ["tables.h", os.path.join(ROOT, "include", "jpeg_host.h")]
LIB_HOST = os.path.join(HERE, "libjpeg_host.so")
CLI = os.path.join(HERE, "jpeg-encoder")
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _abs(p):
    return p if os.path.isabs(p) else os.path.join(CSRC, p)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(_abs(d)) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout)
        raise RuntimeError("build step failed: " + os.path.basename(cmd[-1]))
    return r.stdout


def build_hip(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    objs = []
    for src in HIP_SOURCES:
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + HIP_HEADERS + [os.path.abspath(__file__)]):
            jobs.append([HIPCC] + HIP_FLAGS + ["-c", _abs(src), "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out:
                    print(out)
    if jobs or not os.path.exists(LIB_HIP):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_HIP] + objs)
    return LIB_HIP


def build_host(force=False):
    srcs = [s for s in HOST_SOURCES if os.path.exists(_abs(s))]
    if not srcs:
        return None
    if force or _stale(LIB_HOST, srcs + HOST_HEADERS + [os.path.abspath(__file__)]):
        _run([CXX] + CXX_FLAGS + ["-shared", "-o", LIB_HOST] + [_abs(s) for s in srcs])
    main = _abs("host/main.cpp")
    if os.path.exists(main) and (force or _stale(CLI, srcs + HOST_HEADERS + ["host/main.cpp",
                                                                              os.path.abspath(__file__)])
                                 or os.path.getmtime(LIB_HIP) > os.path.getmtime(CLI)):
        _run([HIPCC, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", CLI, main,
              "-L" + HERE, "-ljpeg_hip", "-ljpeg_host", "-Wl,-rpath,$ORIGIN"])
    return LIB_HOST


def build_all(force=False, verbose=False):
    lib = build_hip(force, verbose)
    build_host(force)
    return lib


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("built", LIB_HIP)
