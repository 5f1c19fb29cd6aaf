"""Build the native pieces in-tree (no JIT cache: the .so travels with the repo snapshot).

  libopencl_wrap_hip.so  = hip_wrap.cpp (C-ABI shim) + whitted_fast.hip + whitted_strict.hip
                           (gfx950 kernels) + scene_prep.c + png_codec.c
Usage: python -m example_gui_opencl_raytracer_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT_DIR = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libopencl_wrap_hip.so")
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CC = os.environ.get("CC") or "gcc"

DEVICE_DEPS = ["whitted_trace.inc", "whitted_hit.inc", "whitted_shade.inc", "whitted_bounce.inc", "whitted_pop.inc", "whitted_tpt.inc", "whitted_terms.inc",
               "whitted_launch.inc", "whitted_params.h"]
UNITS = [
    # (source, compiler, flags, extra deps)
    # -ffp-contract=off on BOTH builds: the fast build fuses where its source says fma, nowhere else.
    # -fno-slp-vectorize: v_pk_{mul,add,fma}_f32 buy no fp32 throughput on gfx950 (plain v_fma_f32 already
    # runs at the vector peak) but cost aligned register pairs: 126 -> 96 VGPRs and -15 % kernel time here.
    ("whitted_fast.hip", "hip", ["-O3", f"--offload-arch={ARCH}", "-fno-slp-vectorize", "-ffp-contract=off"], DEVICE_DEPS),
    ("whitted_strict.hip", "hip", ["-O3", f"--offload-arch={ARCH}", "-fno-slp-vectorize", "-ffp-contract=off"], DEVICE_DEPS),
    ("hip_wrap.cpp", "hip", ["-O2", "-std=c++17", "-Wall"],
     ["whitted_params.h", "scene_prep.h", "png_codec.h", "../../include/opencl_wrap.h", "../../include/hip_wrap_ext.h"]),
    ("scene_prep.c", "c", ["-O2", "-std=c99", "-ffp-contract=off", "-Wall", "-Wextra"], ["scene_prep.h"]),
    ("host_camera.c", "c", ["-O2", "-std=c99", "-ffp-contract=off", "-Wall", "-Wextra", "-D_DEFAULT_SOURCE"],
     ["../../include/hip_wrap_ext.h", "../../include/opencl_wrap.h"]),
    ("png_codec.c", "c", ["-O2", "-std=c99", "-D_DEFAULT_SOURCE", "-pthread", "-Wall", "-Wextra"], ["png_codec.h"]),
]


def kernel_source_hash() -> str:
    """sha256 (first 16 hex digits) over the device sources and their flags: baked into the library (clw_ext_version) and written
    next to every rocprofv3 summary, so that bench.py only quotes profiled counters of the kernels it is timing."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(set(DEVICE_DEPS + ["whitted_fast.hip", "whitted_strict.hip"])):
        h.update(f.encode() + b"\0" + open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(UNITS[0][2]).encode())
    return h.hexdigest()[:16]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, tag: str = "", extra_device_flags=()) -> str:
    """tag / extra_device_flags: side-by-side A/B builds (libopencl_wrap_hip<tag>.so, see tools/)."""
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    lib = LIB if not tag else LIB.replace(".so", f"{tag}.so")
    for src, kind, flags, deps in UNITS:
        spath = os.path.join(CSRC, src)
        opath = os.path.join(OBJ, src.rsplit(".", 1)[0] + tag + ".o")
        if src.endswith(".hip"):
            flags = flags + list(extra_device_flags)
        if src == "hip_wrap.cpp":      # carries the hash of the kernels it is linked with
            flags = flags + [f'-DWT_SOURCE_HASH="{kernel_source_hash()}{tag}"']
            deps = deps + DEVICE_DEPS + ["whitted_fast.hip", "whitted_strict.hip"]
        objs.append(opath)
        dpaths = [spath, __file__] + [os.path.normpath(os.path.join(CSRC, d)) for d in deps]
        if not force and not _newer(opath, dpaths):
            continue
        cmd = ([HIPCC] if kind == "hip" else [CC]) + flags + ["-fPIC", "-c", spath, "-o", opath]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    if force or _newer(lib, objs):
        cmd = [HIPCC, "-shared", "-o", lib] + objs + ["-lz", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    if not tag:      # the pure-C bench driver (host side in the reference's language), linked against the shim
        exe = os.path.join(ROOT_DIR, "tools", "raybench")
        src = os.path.join(ROOT_DIR, "tools", "raybench.c")
        if force or _newer(exe, [src, lib]):
            cmd = [CC, "-O2", "-std=c99", "-D_DEFAULT_SOURCE", "-DCL_TARGET_OPENCL_VERSION=300", "-Wall", "-I", os.path.join(ROOT_DIR, "include"),
                   src, "-o", exe, "-L", HERE, "-lopencl_wrap_hip", "-Wl,-rpath,$ORIGIN/../example_gui_opencl_raytracer_amd",
                   "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
