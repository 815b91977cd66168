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


def _kernel_descriptors(tmp_path, source, extra=()):
    """{kernel name: (private segment bytes, next free VGPR)} of every kernel `source` compiles to for gfx950."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / (os.path.basename(source) + ".s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "--offload-device-only", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                    *extra, source, "-o", str(out)], check=True, cwd=CSRC)
    found = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", out.read_text(), flags=re.S):
        name, body = m.group(1), m.group(2)
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        found[name] = (scratch, vgpr)
    return found


def test_gemm_kernels_use_no_scratch(tmp_path):
    every = _kernel_descriptors(tmp_path, os.path.join(CSRC, "disc.hip"))
    found = {k: v for k, v in every.items() if "disc_gemm" in k}
    assert len(found) >= 8, sorted(found)                      # fp32 engine + register-staged + LDS-DMA instantiations
    spilled = {k: v for k, v in found.items() if v[0] != 0}
    assert not spilled, spilled
    dma = {k: v for k, v in found.items() if "disc_gemm_f16_dma_kernel" in k}
    assert len(dma) >= 5 and all(v[1] <= 256 for v in dma.values()), dma   # 8 waves per workgroup: 256 registers each
    # the fused two-layer kernel (round 4) lives at 254 of 256 registers: one more long-lived address register and it spills
    fused = {k: v for k, v in every.items() if "disc_mlp_fused_kernel" in k}
    assert len(fused) == 6 and all(v[0] == 0 and v[1] <= 256 for v in fused.values()), fused   # KX = 4, 5, 6 x {plane input, raw fp32 rows}


def test_training_step_gemm_kernels_use_no_scratch(tmp_path):
    """disc_train.hip instantiates the LDS-DMA kernel's MODE 2 (plain product, split-K, per-element mask / accumulate
    epilogue) and the fp32-MFMA kernels of the backward pass: same guard -- no private segment, <= 256 registers for the
    8-wave LDS-DMA instantiations."""
    found = {k: v for k, v in _kernel_descriptors(tmp_path, os.path.join(CSRC, "disc_train.hip")).items() if "disc_gemm" in k}
    mode2 = {k: v for k, v in found.items() if "disc_gemm_f16_dma_kernelILi2E" in k}
    assert len(mode2) >= 3, sorted(found)
    assert all(v[0] == 0 for v in found.values()), found
    assert all(v[1] <= 256 for v in mode2.values()), mode2


def test_microbench_kernels_use_no_scratch(tmp_path):
    """tools/gemm_f16_bench.hip (experiment kernels under tools/experiments/): a round-2 build of the panel kernel's 4-wave
    variant kept ~70 registers per lane in scratch and ended in a GPU memory-access fault at 65 536 rows
    (profiles/r02_gemm_f16_l1_panel_experiment.txt); no experiment kernel may carry a private segment onto the GPU box."""
    found = _kernel_descriptors(tmp_path, os.path.join(ROOT, "tools", "gemm_f16_bench.hip"), extra=("-I", CSRC))
    gemms = {k: v for k, v in found.items() if "disc_gemm" in k}   # (the bare-MFMA calibration loops park 8 B: not GEMM kernels)
    assert any("panel" in k for k in gemms) and len(gemms) >= 10, sorted(found)
    spilled = {k: v for k, v in gemms.items() if v[0] != 0}
    assert not spilled, spilled


def test_env_step_dma_kernels_use_no_scratch(tmp_path):
    """The env-step DMA tile body went through scratch twice while it was written (a dynamically indexed by-value struct
    member, a compiler-built pointer table) and reached 320 registers with per-key branches around loads; the guard: no
    private segment and at least five waves per SIMD (<= 96 VGPRs) for every instantiation, the expert body included."""
    found = _kernel_descriptors(tmp_path, os.path.join(CSRC, "env_step.hip"))
    dma = {k: v for k, v in found.items() if "env_step_dma" in k}
    assert len(dma) == 6, sorted(found)                         # {plain, fused with the expert sample} x tile {32, 16, 8}
    assert all(v[0] == 0 for v in found.values()), found
    assert all(v[1] <= 96 for v in dma.values()), dma


def test_reset_and_scatter_kernels_use_no_scratch(tmp_path):
    """The one-launch device reset (motion.hip) and the row scatter (compact.hip, a dynamically indexed by-value op table)."""
    for src in ("motion.hip", "compact.hip", "command.hip"):
        found = _kernel_descriptors(tmp_path, os.path.join(CSRC, src))
        assert found and all(v[0] == 0 for v in found.values()), (src, found)
