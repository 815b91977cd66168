"""CPU: the hand-scheduled GEMM kernels must not spill.  hipcc cross-compiles disc.hip for gfx950 to assembly and the
kernel descriptors are read back: every discriminator GEMM kernel has a zero private segment (no scratch).  A harmless-
looking `break` in the LDS-DMA kernel's k-loop once cost its 256 x 256 instantiations 350-420 B of spills and 7x the time
while every parity test stayed green -- this test is the guard for that class of regression."""

import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "humanoid_amp_amd", "csrc")


def test_gemm_kernels_use_no_scratch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "disc.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "--offload-device-only", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                    os.path.join(CSRC, "disc.hip"), "-o", str(out)], check=True, cwd=CSRC)
    text = out.read_text()
    found = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, flags=re.S):
        name, body = m.group(1), m.group(2)
        if "disc_gemm" not in name:
            continue
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        found[name] = (scratch, vgpr)
    assert len(found) >= 8, sorted(found)                      # fp32 engine + register-staged + LDS-DMA instantiations
    spilled = {k: v for k, v in found.items() if v[0] != 0}
    assert not spilled, spilled
    dma = {k: v for k, v in found.items() if "disc_gemm_f16_dma_kernel" in k}
    assert len(dma) >= 5 and all(v[1] <= 256 for v in dma.values()), dma   # 8 waves per workgroup: 256 registers each


def test_env_step_dma_kernels_use_no_scratch(tmp_path):
    """The env-step DMA tile body went through scratch twice while it was written (a dynamically indexed by-value struct
    member, a compiler-built pointer table) and reached 320 registers with per-key branches around loads; the guard: no
    private segment and at least five waves per SIMD (<= 96 VGPRs) for every instantiation, the expert body included."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "env_step.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "--offload-device-only", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                    os.path.join(CSRC, "env_step.hip"), "-o", str(out)], check=True, cwd=CSRC)
    found = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", out.read_text(), flags=re.S):
        name, body = m.group(1), m.group(2)
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        found[name] = (scratch, vgpr)
    dma = {k: v for k, v in found.items() if "env_step_dma" in k}
    assert len(dma) == 6, sorted(found)                         # {plain, fused with the expert sample} x tile {32, 16, 8}
    assert all(v[0] == 0 for v in found.values()), found
    assert all(v[1] <= 96 for v in dma.values()), dma
