"""Build recipe of ``csrc/libamp_engine.so`` (hipcc, gfx950 only, in-tree so the .so travels with the repo)."""

from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["core.hip", "motion.hip", "env_step.hip", "compact.hip", "disc.hip", "disc_train.hip", "ring.hip", "convert.hip"]
HEADERS = ["amp_common.hpp", "disc_gemm.hpp", "disc_gemm_f16.hpp", "disc_gemm_f16_dma.hpp", os.path.join("..", "..", "include", "amp_engine.h")]
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
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source into libamp_engine.so; returns its path."""
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc()] + FLAGS + ["-o", LIB + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[humanoid_amp_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True))
