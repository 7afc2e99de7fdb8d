#!/usr/bin/env python3
"""Build the native pieces in-tree (no install step; the .so files travel with the repo).

    lib/libaz_mcts.so          HIP engine + C ABI (include/az_mcts.h), hipcc --offload-arch=gfx950
    src/mcts_cpp.<ext>.so      CPython extension over the C ABI (reference module name/location,
                               setup.py:36-65 of the reference: `from src import mcts_cpp`)
    src/env_cpp.<ext>.so       CPython extension with the host-side Env objects

Device code is compiled with -ffp-contract=off: the search must round exactly like the
reference (explicit fmaf() marks the three contractions its compiled form contains).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "lib")
SRC = os.path.join(HERE, "src")
EXT = sysconfig.get_config_var("EXT_SUFFIX")
ARCH = os.environ.get("AZ_OFFLOAD_ARCH", "gfx950")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _headers():
    """Every header a native piece may include: editing one (the game rules in games.h, the record
    layout, the C ABI) rebuilds the engine AND the modules."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INC, "*.h")))


def _run(cmd):
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


# nn_stem, nn_heads: MFMA results in VGPRs (their vector code reads every accumulator element right away; from AGPRs
# that is one v_accvgpr_read each; the other kernels get VGPR accumulators anyway)
PER_FILE_FLAGS = {"nn_stem": ("-mllvm", "-amdgpu-mfma-vgpr-form=1"), "nn_heads": ("-mllvm", "-amdgpu-mfma-vgpr-form=1")}


def build_engine(force=False):
    """One object per .hip source (compiled in parallel, rebuilt only when that source or a header
    changed), linked into lib/libaz_mcts.so.  No device code crosses translation units."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(LIB, "libaz_mcts.so")
    names = ("kernels", "tt_kernels", "engine", "nn_kernels", "nn_conv", "nn_conv2", "nn_stem", "nn_attn", "nn_heads", "nn_model", "nn_othello", "nn_othello_heads")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = _headers()
    jobs = []
    for n in names:
        src, obj = os.path.join(CSRC, n + ".hip"), os.path.join(objdir, n + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                         *os.environ.get("AZ_EXTRA_CFLAGS", "").split(),      # experiments only (e.g. -DAZ_OTH_EXPERIMENTS)
                         *PER_FILE_FLAGS.get(n, ()),
                         "-I", INC, "-I", CSRC, "-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            list(pool.map(_run, jobs))
    objs = [os.path.join(objdir, n + ".o") for n in names]
    if force or jobs or _stale(out, objs):
        _run([hipcc, f"--offload-arch={ARCH}", "-fPIC", "-shared", *objs, "-o", out])
    return out


def build_module(name, source, link_engine, force=False):
    import pybind11
    os.makedirs(SRC, exist_ok=True)
    out = os.path.join(SRC, name + EXT)
    src = os.path.join(CSRC, source)
    if force or _stale(out, [src] + _headers()):
        cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
               "-I", INC, "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"],
               src, "-o", out]
        if link_engine:
            cmd += ["-L", LIB, "-laz_mcts", "-Wl,-rpath,$ORIGIN/../lib"]
        _run(cmd)
    return out


def build_all(force=False):
    build_engine(force)
    build_module("mcts_cpp", "mcts_module.cpp", True, force)
    build_module("env_cpp", "env_module.cpp", False, force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
