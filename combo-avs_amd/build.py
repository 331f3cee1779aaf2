#!/usr/bin/env python3
"""Builds libcombo_avs_hip.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python combo-avs_amd/build.py [--force] [--save-temps]

hipcc cross-compiles without a GPU.  The .so lands in combo-avs_amd/lib/ (git-ignored, but it travels
with gpurun snapshots).  One translation unit per .hip file, compiled in parallel, then linked.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libcombo_avs_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wno-unused-result"]


def _newer(a, bs):
    if not os.path.exists(a):
        return False
    t = os.path.getmtime(a)
    return all(os.path.getmtime(b) <= t for b in bs)


def build(force=False, save_temps=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "combo_avs.h")]
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not _newer(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + (["-save-temps=obj"] if save_temps else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJDIR)
        return cmd, r
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for cmd, r in ex.map(run, jobs):
            if verbose:
                print("[build]", os.path.basename(cmd[-3]), "ok" if r.returncode == 0 else "FAILED")
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError("hipcc failed: " + " ".join(cmd))
            if r.stderr.strip() and verbose:
                sys.stderr.write(r.stderr)
    if jobs or not os.path.exists(LIB) or force:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print("[build] linked", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
