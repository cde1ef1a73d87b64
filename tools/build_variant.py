#!/usr/bin/env python3
"""Build the HIP library of the working tree under another name, optionally with extra compiler flags, for A/B
timings on one GPU box (tools/ab_bench.sh) and for instrumented builds:

    python tools/build_variant.py <name> [extra hipcc flags...]   ->  jpeg-encoder_amd/variants/<name>.so
"""
import importlib.util
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("jpg_build", os.path.join(ROOT, "jpeg-encoder_amd", "build.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(B.HERE, "variants")
    obj_dir = os.path.join(out_dir, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    objs, jobs = [], []
    for src in B.HIP_SOURCES:
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        jobs.append([B.HIPCC] + B.HIP_FLAGS + extra + ["-c", B._abs(src), "-o", obj])
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(B._run, jobs))
    lib = os.path.join(out_dir, name + ".so")
    B._run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    subprocess.run(["rm", "-rf", obj_dir])
    print("built", lib)


if __name__ == "__main__":
    main()
