"""Build recipe of ``csrc/libamp_engine.so`` (hipcc, gfx950 only, in-tree so the .so travels with the repo)."""

from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["core.hip", "motion.hip", "env_step.hip", "compact.hip", "command.hip", "disc.hip", "disc_train.hip", "ring.hip", "convert.hip", "hot_step.hip", "calibrate.hip"]



def _headers():
    """Every header a source may include: all of csrc/*.hpp (globbed, so a new header cannot be forgotten) + the ABI."""
    import glob

    return sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(HERE, "..", "include", "amp_engine.h")]


LIB = os.path.join(CSRC, "libamp_engine.so")
# -ffp-contract=off: the reference's fp32 op order (separate mul/sub in sqrt(1 - c*c), rounded quaternion dot)
# is part of the parity contract; never -ffast-math.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (expected on PATH or at /opt/rocm/bin/hipcc)")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers() + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source into libamp_engine.so; returns its path."""
    if not force and not is_stale():
        return LIB
    # one object per source, compiled in parallel (objects live under csrc/build/, git-ignored), then one link
    from concurrent.futures import ThreadPoolExecutor

    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + ["-c"]
    hdr_time = max(os.path.getmtime(h) for h in _headers() + [os.path.abspath(__file__)])

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
            return obj
        cmd = [_hipcc()] + cflags + ["-o", obj, path]
        if verbose:
            print("[humanoid_amp_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    if verbose:
        print("[humanoid_amp_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True))
