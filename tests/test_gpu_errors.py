"""GPU: argument validation at the boundary (error behaviour mirrors the reference's exception types) and edge shapes."""

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def test_motion_loader_error_types():
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.motions import MotionLoader

    with pytest.raises(ValueError, match="No files found"):
        MotionLoader("/nope/missing.npz", "cuda:0")
    ml = MotionLoader(gu.clip_files("g1_walk")[0], "cuda:0")
    with pytest.raises(AssertionError):
        ml.get_body_index(["torso"])  # humanoid name on a G1 clip
    with pytest.raises(nat.AmpEngineError, match="set_obs_layout"):
        ml.collect_reference(np.zeros(2), np.zeros(2, dtype=np.int64), 2)
    with pytest.raises(nat.AmpEngineError):
        ml.set_obs_layout(list(range(29)), 99, [7, 8, 9, 10])  # reference body out of range
    with pytest.raises(ValueError):
        ml.set_obs_layout(list(range(5)), 0, [7, 8, 9, 10])
    ml.set_obs_layout(list(range(29)), 0, [7, 8, 9, 10])
    out = ml.collect_reference(np.zeros(0), np.zeros(0, dtype=np.int64), 2)
    assert out.shape == (0, 166)
    # K = 1 (no history) is legal
    one = ml.collect_reference(np.array([0.3]), np.array([0]), 1)
    two = ml.collect_reference(np.array([0.3]), np.array([0]), 2)
    assert torch.equal(one[0], two[0, :83])


def test_env_step_argument_checks():
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import EnvStepConfig, EnvStepKernel

    ker = EnvStepKernel(EnvStepConfig(n_dof=29, num_amp_observations=2, max_episode_length=300, rew_track_vel=1.0), 100, "cuda:0")
    z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    with pytest.raises(nat.AmpEngineError, match="null buffer"):
        ker.launch(nat.AMP_PHASE_OBS, joint_pos=z(100, 29))
    with pytest.raises(nat.AmpEngineError, match="rows"):
        ker.launch(nat.AMP_PHASE_DONES, root_pos=z(99, 3), episode_length=torch.zeros(100, dtype=torch.long, device="cuda"))
    with pytest.raises(nat.AmpEngineError, match="contiguous last dim"):
        ker.launch(nat.AMP_PHASE_DONES, root_pos=z(100, 6)[:, ::2], episode_length=torch.zeros(100, dtype=torch.long, device="cuda"))
    with pytest.raises(nat.AmpEngineError, match="phases"):
        ker.launch(0)
    # strided Isaac-style views are accepted
    body = z(100, 39, 3)
    ker.launch(nat.AMP_PHASE_DONES, root_pos=body[:, 0], episode_length=torch.full((100,), 299, dtype=torch.long, device="cuda"))
    assert bool(ker.time_out.all()) and bool(ker.died.all())  # z = 0 < 0.5 and ep_len >= max - 1
    ids, count = ker.compact_resets()
    assert int(count) == 100 and torch.equal(ids[:100].cpu(), torch.arange(100))


def test_discriminator_argument_checks():
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import AmpDiscriminator
    from oracle import disc as odisc

    w = odisc.make_weights(166, seed=0)
    with pytest.raises(nat.AmpEngineError, match="multiple of"):
        AmpDiscriminator([(torch.zeros(1000, 166), torch.zeros(1000)), (torch.zeros(512, 1000), torch.zeros(512)),
                          (torch.zeros(1, 512), torch.zeros(1))], "cuda:0")
    with pytest.raises(ValueError, match="inconsistent"):
        AmpDiscriminator([w[0], (torch.zeros(512, 1000), torch.zeros(512)), w[2]], "cuda:0")
    d = AmpDiscriminator(w, "cuda:0")
    with pytest.raises(nat.AmpEngineError, match="float32"):
        d.style_reward(torch.zeros(4, 165, device="cuda"))
    with pytest.raises(ValueError, match="entries"):
        d.set_scaler(torch.zeros(10, dtype=torch.float64), torch.ones(10, dtype=torch.float64))
    assert d.style_reward(torch.zeros(0, 166, device="cuda"))["style"].shape == (0, 1)
    # a row-strided view (every other row of a wider buffer) is accepted without a copy
    big = torch.randn(64, 2 * 166, device="cuda")
    a = d.style_reward(big[:, :166])["style"]
    b = d.style_reward(big[:, :166].contiguous())["style"]
    assert torch.equal(a, b)


def test_mfma_calibration_streams_and_argument_checks():
    """amp_calibrate_mfma_f16 (the measurement aid behind bench.py's roofline.mfma_sustained_tflops): both streams run and land in a
    plausible band of the nominal 2 516.8 TFLOP/s; the layer-2 stream (v_mfma_f32_16x16x32_f16, two waves per SIMD) needs 512 floats
    of scratch per CU and says so."""
    import ctypes as C

    from humanoid_amp_amd import _native as nat

    legacy, mhz = nat.calibrate_mfma_f16(True, 64, with_clock=True)
    layer2, mhz2 = nat.calibrate_mfma_f16(True, 64, with_clock=True, layer2_stream=True)
    const2 = nat.calibrate_mfma_f16(False, 64, layer2_stream=True)
    for v in (legacy, layer2, const2):
        assert 500.0 < v < 2600.0, v
    assert 800.0 < mhz < 2600.0 and 800.0 < mhz2 < 2600.0
    assert const2 > layer2  # constant operands: no multiplier toggling, higher clock
    lib = nat.load()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    small = torch.zeros(cus * 256, device="cuda")
    flops = C.c_double()
    with torch.cuda.device(0):
        rc = lib.amp_calibrate_mfma_f16(3, 8, nat.dptr(small), small.numel(), C.byref(flops), nat.stream_ptr())
    assert rc < 0 and b"scratch must hold" in lib.amp_last_error()
