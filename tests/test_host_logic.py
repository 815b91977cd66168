"""CPU: host-side logic of the drop-in classes (no kernels run): path mini-language, name lookups, the numpy RNG of
sample_times, config sizes, sharding arithmetic, loud failure without a GPU, and the oracle-isolation rule."""

import ast
import os

import numpy as np
import pytest
import torch
import yaml

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_resolver_mini_language(tmp_path):
    from humanoid_amp_amd.motions.motion_loader import _resolve_motion_files as res

    a, b = gu.clip_files("humanoid3")[:2]
    assert res(a) == [a]
    assert res(f"{a}, {b}") == [a, b]
    assert res(os.path.join(gu.MOTIONS, "humanoid_*.npz")) == sorted(gu.clip_files("humanoid3"))
    assert len(res(gu.MOTIONS)) == 6  # directory -> every npz, sorted (five BASELINE clips + custom_motion.npz)
    cfg = tmp_path / "m.yaml"
    cfg.write_text(yaml.safe_dump({"motion_files": [a, "missing.npz"]}))
    assert res(str(cfg)) == [a]
    cfg.write_text(yaml.safe_dump({"glob_pattern": os.path.join(gu.MOTIONS, "G1_*.npz")}))
    assert [os.path.basename(p) for p in res(str(cfg))] == ["G1_dance.npz", "G1_walk.npz"]
    cfg.write_text(yaml.safe_dump({"motion_files": ["nope.npz"]}))
    with pytest.raises(ValueError, match="No valid motion files found in config"):
        res(str(cfg))
    with pytest.raises(ValueError, match="No files found for pattern"):
        res("/definitely/not/here.npz")
    with pytest.raises(ValueError, match="No files found for pattern"):
        res("/nope/*.npz")


def test_loader_metadata_and_name_lookup(capsys):
    from humanoid_amp_amd.motions import MotionLoader

    ml = MotionLoader(",".join(gu.clip_files("humanoid3")), "cpu")
    out = capsys.readouterr().out
    assert "Loading 3 motion file(s) from:" in out and "Motion loaded: 3 files, total duration:" in out
    fx = gu.golden("frame_blend_humanoid3")
    assert ml.num_trajectories == 3 and ml.num_frames == 1138 and ml.num_dofs == 28 and ml.num_bodies == 15
    assert float(ml.dt) == float(fx["dt"]) and np.array_equal(ml.durations, fx["durations"])
    assert np.array_equal(ml.traj_starts, fx["traj_starts"]) and np.array_equal(ml.traj_ends, fx["traj_ends"])
    assert ml.duration == float(np.sum(fx["durations"]))
    assert ml.dof_positions.dtype == torch.float32 and ml.body_rotations.shape == (1138, 15, 4)
    assert ml.get_body_index(["torso", "right_hand"]) == [1, 5]
    with pytest.raises(AssertionError, match="doesn't exist"):
        ml.get_dof_index(["not_a_joint"])
    with pytest.raises(AssertionError, match="doesn't exist"):
        ml.get_body_index(["not_a_body"])


def test_sample_times_uses_the_global_numpy_rng_like_the_reference():
    from humanoid_amp_amd.motions import MotionLoader

    ml = MotionLoader(",".join(gu.clip_files("humanoid3")), "cpu")
    np.random.seed(123)
    ids, t = ml.sample_times(1000)
    np.random.seed(123)  # motions/motion_loader.py:321-327: randint first, then uniform * durations[ids]
    want_ids = np.random.randint(0, 3, size=1000)
    want_t = np.random.uniform(low=0.0, high=1.0, size=1000) * ml.durations[want_ids]
    assert np.array_equal(ids, want_ids) and np.array_equal(t, want_t)
    ids0, t0 = ml.sample_times(5, start=True)
    assert t0.dtype == np.float64 and not t0.any() and ids0.shape == (5,)


def test_no_cpu_fallback():
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import AmpDiscriminator, EnvStepConfig, EnvStepKernel, reset_compact
    from humanoid_amp_amd.motions import MotionLoader

    ml = MotionLoader(gu.clip_files("g1_walk")[0], "cpu")
    with pytest.raises(nat.AmpEngineError, match="no CPU fallback"):
        ml.sample(2, times=np.zeros(2))
    with pytest.raises(nat.AmpEngineError):
        ml._compute_frame_blend(np.zeros(2), np.zeros(2, dtype=np.int64))
    with pytest.raises(nat.AmpEngineError):
        EnvStepKernel(EnvStepConfig(n_dof=29, num_amp_observations=2, max_episode_length=300), 8, "cpu")
    with pytest.raises(nat.AmpEngineError):
        reset_compact(torch.zeros(8, dtype=torch.bool))
    with pytest.raises(nat.AmpEngineError):
        AmpDiscriminator([(torch.zeros(1024, 166), torch.zeros(1024)), (torch.zeros(512, 1024), torch.zeros(512)),
                          (torch.zeros(1, 512), torch.zeros(1))], "cpu")


def test_missing_library_is_an_import_error(monkeypatch, tmp_path):
    from humanoid_amp_amd import _native as nat

    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", str(tmp_path / "libamp_engine.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        nat.load()


def test_policy_obs_sizes_match_the_reference_configs():
    """observation_space arithmetic of g1_amp_env_cfg.py:186-206, through the C ABI (host-only call)."""
    import ctypes as C

    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import EnvStepConfig
    from humanoid_amp_amd.envs import G1AmpDeployEnvCfg

    lib = nat.load()
    size = lambda **kw: int(lib.amp_policy_obs_size(C.byref(EnvStepConfig(n_dof=29, num_amp_observations=2,  # noqa: E731
                                                                              max_episode_length=300, **kw).to_c())))
    assert size(rew_track_vel=1.0) == 102 and size() == 100
    assert size(rew_track_vel=1.0, num_actor_observations=3) == 306 == int(gu.golden("meta")["deploy_hist3_observation_space"])
    assert size(rew_track_vel=1.0, num_actor_observations=2, history_include_last_actions=False,
                history_include_command=False) == 173
    assert G1AmpDeployEnvCfg().observation_space == size(rew_track_vel=1.0, num_actor_observations=2)
    hum = EnvStepConfig(n_dof=28, num_amp_observations=2, max_episode_length=300, use_last_actions=False, reward_mode=0)
    assert int(lib.amp_policy_obs_size(C.byref(hum.to_c()))) == 81


def test_task_table_and_cfg_values():
    from humanoid_amp_amd.envs import TASKS, G1AmpDanceEnvCfg, HumanoidAmpWalkEnvCfg

    assert len(TASKS) == 7 and "Isaac-G1-AMP-Dance-Direct-v0" in TASKS
    c = G1AmpDanceEnvCfg()
    assert (c.num_amp_observations, c.amp_observation_space, c.observation_space, c.decimation) == (10, 83, 102, 1)
    assert (c.rew_termination, c.rew_action_l2, c.rew_joint_pos_limits, c.rew_joint_acc_l2, c.rew_joint_vel_l2, c.rew_track_vel) \
        == (-1.0, -0.1, -10, -1.0e-06, -0.001, 1.0)
    assert c.reset_strategy == "random-start" and os.path.basename(c.motion_file) == "G1_dance.npz"
    h = HumanoidAmpWalkEnvCfg()
    assert (h.num_amp_observations, h.amp_observation_space, h.reference_body, h.scene.env_spacing) == (2, 81, "torso", 10.0)


def test_workload_accounting_matches_the_survey():
    from humanoid_amp_amd.workloads import WORKLOADS, algorithmic_bytes_per_env_step, disc_flops_per_row

    assert algorithmic_bytes_per_env_step(WORKLOADS["g1_walk"]) == 2682
    assert algorithmic_bytes_per_env_step(WORKLOADS["g1_dance"]) == 10650
    assert disc_flops_per_row(166) == 1389568 and disc_flops_per_row(830) == 2749440 and disc_flops_per_row(162) == 1381376


def test_shard_bounds_cover_every_env_once():
    from humanoid_amp_amd.distributed import global_env_ids, shard_bounds

    for n, w in ((65536, 8), (32768, 4), (10, 3), (7, 8)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    assert global_env_ids(torch.tensor([0, 5]), 65536, 8, 3).tolist() == [24576, 24581]
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    import itertools

    walk = itertools.chain(os.walk(os.path.join(ROOT, "humanoid_amp_amd")), os.walk(os.path.join(ROOT, "tools")))
    for dirpath, _, files in walk:
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom) and node.level == 0:
                    mods = [node.module or ""]
                assert not any(m == "oracle" or m.startswith("oracle.") for m in mods), os.path.join(dirpath, f)
    # bench.py: the oracle appears only inside cpu_baseline()
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(fn))
        assert uses == (fn.name == "cpu_baseline"), fn.name


def test_shipped_clips_pass_the_surveys_self_checks():
    """the only self-checks the reference's own data offers -- fps = 60 in every clip, the clips' frame
    counts, unit body quaternions -- hold for the clips this package ships (read with allow_pickle=False)."""
    import glob

    import numpy as np

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "humanoid_amp_amd", "motions")
    want = {"G1_walk.npz": (399, 29, 11), "G1_dance.npz": (601, 29, 39), "humanoid_dance.npz": (902, 28, 15),
            "humanoid_run.npz": (82, 28, 15), "humanoid_walk.npz": (154, 28, 15), "custom_motion.npz": (309, 29, 25)}
    seen = {}
    for path in sorted(glob.glob(os.path.join(root, "*.npz"))):
        d = np.load(path, allow_pickle=False)
        assert int(d["fps"]) == 60, path
        frames, dofs = d["dof_positions"].shape
        bodies = d["body_rotations"].shape[1]
        seen[os.path.basename(path)] = (frames, dofs, bodies)
        for key in ("dof_velocities", "body_positions", "body_linear_velocities", "body_angular_velocities"):
            assert d[key].shape[0] == frames and d[key].dtype in (np.float32, np.float64), (path, key)
        norm = np.linalg.norm(d["body_rotations"].astype(np.float64), axis=-1)
        assert norm.min() > 1.0 - 1e-6 and norm.max() < 1.0 + 1e-6, (path, norm.min(), norm.max())
        assert len(d["dof_names"]) == dofs and len(d["body_names"]) == bodies
    assert seen == want


def test_tt_gemm_shape_admission():
    """The weight-gradient ("TT") GEMM kernel has no row / column guards: the host plan admits only shapes its tile covers
    exactly.  Round 3 checked `% 64` whatever the tile, so a 128-wide tile on N = kN = 192 (K D = 166) or 832 (K D = 830)
    would have read and stored 64 columns past every row (profiles/r04_tt_kernel_abort.md).  Host-only call."""
    import ctypes as C

    from humanoid_amp_amd import _native as nat

    lib = nat.load()

    def plan(M, N, K, lda=None, ldw=None, ldc=None, split=1):
        bm, bn, sl = C.c_int32(), C.c_int32(), C.c_int32()
        rc = lib.amp_disc_train_tt_plan(M, N, K, lda or M, ldw or N, ldc or N, split, C.byref(bm), C.byref(bn), C.byref(sl))
        return None if rc != 0 else (bm.value, bn.value, sl.value)

    # every product of the training step at BASELINE's minibatch (3 x 4096 rows; gradient-penalty products reduce over 4096):
    # gW2 [512 x 1024], gW1 [1024 x kN] with kN = 192 (K D = 166) / 832 (K D = 830) -- multiples of 64, NOT of 128
    for (M, N, K) in [(512, 1024, 12288), (1024, 192, 12288), (1024, 832, 12288), (512, 1024, 4096), (1024, 192, 4096),
                      (1024, 832, 4096), (512, 1024, 2304), (1024, 192, 6), (64, 64, 1)]:
        p = plan(M, N, K)
        assert p is not None, (M, N, K)
        bm, bn, sl = p
        assert M % bm == 0 and N % bn == 0 and bm in (64, 128) and bn in (64, 128), (M, N, K, p)   # the tile covers the shape exactly
        assert 1 <= sl <= 16 and (sl & (sl - 1)) == 0
        assert sl == 1 or (K + 15) // 16 // sl >= 16                                                # >= 16 k-tiles per slice
    assert plan(1024, 192, 12288)[1] == 64 and plan(1024, 832, 12288)[1] == 64                      # never a 128-wide tile on kN
    assert plan(1024, 832, 12288)[0] == 128                                                         # (rows do divide: 128 x 64)
    assert plan(1024, 832, 12288, split=0)[2] == 1
    # the slice count does not depend on the tile: a row of C is the same sum whatever tile ran it
    assert plan(1024, 832, 12288)[2] == plan(1024, 832 - 64, 12288)[2] == 8
    # refused: ragged tiles, rows that do not hold the tile, unaligned pitches, empty / oversized reductions
    for bad in [dict(M=1000, N=192, K=4096), dict(M=1024, N=166, K=4096), dict(M=1024, N=192, K=0), dict(M=1024, N=192, K=2**31),
                dict(M=1024, N=192, K=4096, lda=1000), dict(M=1024, N=192, K=4096, ldw=166), dict(M=1024, N=192, K=4096, ldc=128),
                dict(M=1024, N=192, K=4096, ldw=194), dict(M=0, N=192, K=4096), dict(M=-64, N=192, K=4096)]:
        assert plan(**bad) is None, bad
    assert b"not admitted" in lib.amp_last_error()


def test_engine_library_can_be_selected_by_environment(tmp_path):
    """AMP_ENGINE_LIB (A/B variants must not overwrite the in-tree product library): the binding loads the named file,
    still checks every symbol and the ABI version, and falls back to the in-tree library when the variable is unset or empty."""
    import shutil
    import subprocess
    import sys

    from humanoid_amp_amd import _native as nat

    variant = tmp_path / "libamp_variant.so"
    shutil.copy(nat.LIB_PATH if not os.environ.get("AMP_ENGINE_LIB") else os.path.join(ROOT, "humanoid_amp_amd", "csrc", "libamp_engine.so"),
                variant)
    code = "from humanoid_amp_amd import _native as n; n.load(); print(n.LIB_PATH, n.load().amp_abi_version())"
    for env_value, want in ((str(variant), str(variant)), ("", os.path.join("csrc", "libamp_engine.so"))):
        env = dict(os.environ, AMP_ENGINE_LIB=env_value, PYTHONPATH=ROOT)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
        assert out[0].endswith(want) and int(out[1]) == nat.ABI_VERSION
    bad = tmp_path / "missing.so"
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AMP_ENGINE_LIB=str(bad), PYTHONPATH=ROOT), capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_step_counter_hand_over_arguments_are_checked_on_the_host():
    """ABI 10: AmpPrePhysicsArgs.step_in / step_out hand the device-side step counter of a captured env step to the reset
    launch.  The pair comes together, names two different words, and the tick of the same launch must not read the word the
    launch writes -- all refused on the host, before anything is launched (no GPU needed for the refusals)."""
    import ctypes as C

    from humanoid_amp_amd import _native as nat

    lib = nat.load()
    words = (C.c_uint64 * 2)(7, 0)
    base = C.addressof(words)
    buf = (C.c_float * 64)()

    def args(step_in, step_out):
        a = nat.AmpPrePhysicsArgs()
        a.actions_in = C.addressof(buf)
        a.num_envs, a.n_actions = 4, 3
        a.step_in, a.step_out = step_in, step_out
        return a

    for step_in, step_out in ((base, None), (None, base + 8), (base, base)):
        assert lib.amp_pre_physics_step(C.byref(args(step_in, step_out)), None, None) != 0
        assert b"step_in / step_out" in lib.amp_last_error()
    tick = nat.AmpCommandArgs()
    tick.command, tick.time_left = C.addressof(buf), C.addressof(buf)
    tick.step_dev = base + 8   # the word this launch would write
    assert lib.amp_pre_physics_step(C.byref(args(base, base + 8)), C.byref(tick), None) != 0
    assert b"must not read the word" in lib.amp_last_error()
