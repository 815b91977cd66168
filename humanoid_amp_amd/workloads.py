"""The BASELINE.json workloads as runnable objects: one env-step of the AMP hot path on one env shard.

One env-step, per env:
  1. motion-sample: (t, clip) -> K expert frames -> expert AMP obs [K*D]        (amp_collect_reference)
  2. sim AMP obs + K-history shift + policy obs                                  (amp_env_step, OBS)
  3. done mask + ascending reset-id compaction                                   (amp_env_step DONES + amp_reset_compact_tiles)
  4. task reward (G1; humanoid: constant 1)                                      (amp_env_step, REWARD)
  5. scaler + discriminator MLP + style reward + reward mix                      (amp_disc_style_reward)
Physics is excluded (closed PhysX step in the reference).
"""

from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as nat
from .engine import AmpDiscriminator, EnvStepConfig, EnvStepKernel
from .motions import MOTIONS_DIR, MotionLoader
from .robots import G1_23DOF_DROPPED, G1_23DOF_JOINT_NAMES, G1_JOINT_NAMES, G1_KEY_BODY_NAMES, HUMANOID_KEY_BODY_NAMES
from .synthetic import make_state

# reward scales of the self-consistent G1 config family (g1_amp_env_cfg.py:86-91)
G1_REWARDS = dict(rew_termination=-1.0, rew_action_l2=-0.1, rew_joint_pos_limits=-10.0, rew_joint_acc_l2=-1.0e-06,
                  rew_joint_vel_l2=-0.001, rew_track_vel=1.0)


@dataclass(frozen=True)
class WorkloadSpec:
    name: str
    description: str
    clips: tuple
    n_dof: int
    K: int
    robot: str               # "g1" | "humanoid"
    reference_body: str
    episode_length_s: float
    decimation: int
    task_weight: float       # agents/*.yaml task_reward_weight
    style_weight: float
    drop_dofs: tuple = ()    # synthetic variant: the clips WITHOUT these DoF columns (derived files, see clip_files)

    @property
    def joint_names(self):
        """Robot-side joint order (g1 only; the humanoid takes the clip's own order)."""
        return G1_23DOF_JOINT_NAMES if self.drop_dofs else G1_JOINT_NAMES

    @property
    def D(self) -> int:
        return 2 * self.n_dof + 25

    @property
    def max_episode_length(self) -> int:
        import math

        return math.ceil(self.episode_length_s / ((1.0 / 60.0) * self.decimation))


WORKLOADS = {
    # BASELINE.json configs[1] / configs[4]: "G1-AMP-Walk" (29 DoF in every clip; "23-DoF" is prose only)
    "g1_walk": WorkloadSpec("g1_walk", "G1-AMP-Walk, G1_walk.npz (399 frames, 11 bodies), 29-DoF, K=2, D=83", ("G1_walk",),
                            29, 2, "g1", "pelvis", 10.0, 2, 0.0, 1.0),
    # configs[2]: long multi-phase clip, K = 10
    # configs[1] read literally ("23-DoF"): a SYNTHETIC variant -- G1_walk.npz without the six DoF columns the 23-DoF G1 lacks
    # (D = 71, K D = 142, policy obs 84); the derived clip is written next to the system's temp files on first use
    "g1_walk_23dof": WorkloadSpec("g1_walk_23dof", "G1-AMP-Walk, synthetic 23-DoF variant of G1_walk.npz (399 frames, 11 bodies; waist roll / "
                                  "pitch and wrist pitch / yaw columns dropped), K=2, D=71", ("G1_walk",), 23, 2, "g1", "pelvis", 10.0, 2,
                                  0.0, 1.0, drop_dofs=G1_23DOF_DROPPED),
    "g1_dance": WorkloadSpec("g1_dance", "G1-AMP-Dance, G1_dance.npz (601 frames, 39 bodies), 29-DoF, K=10, D=83", ("G1_dance",),
                             29, 10, "g1", "pelvis", 10.0, 1, 1.0, 1.0),
    # configs[3]: walk+run+dance multi-clip blend, 28 DoF
    "humanoid3": WorkloadSpec("humanoid3", "Humanoid-AMP walk+run+dance 3-clip table (1138 frames, 15 bodies), 28-DoF, K=2, D=81",
                              ("humanoid_walk", "humanoid_run", "humanoid_dance"), 28, 2, "humanoid", "torso", 10.0, 2, 0.0, 1.0),
}


def clip_files(spec: WorkloadSpec) -> list:
    """The npz files of a workload's clips.  A synthetic variant (``drop_dofs``) gets DERIVED clips: the source arrays with those DoF
    columns removed (``dof_names``, ``dof_positions``, ``dof_velocities``; bodies untouched), written once per process tree into the
    temp directory -- data derived from the shipped clips, nothing fetched."""
    src = [os.path.join(MOTIONS_DIR, c + ".npz") for c in spec.clips]
    if not spec.drop_dofs:
        return src
    import tempfile

    out_dir = os.path.join(tempfile.gettempdir(), "humanoid_amp_amd_clips")
    os.makedirs(out_dir, exist_ok=True)
    out = []
    for path in src:
        dst = os.path.join(out_dir, f"{os.path.basename(path)[:-4]}_{spec.n_dof}dof.npz")
        if not os.path.exists(dst):
            with np.load(path, allow_pickle=False) as d:
                arrays = {k: d[k] for k in d.files}
            keep = [i for i, n in enumerate(arrays["dof_names"].tolist()) if n not in spec.drop_dofs]
            assert len(keep) == spec.n_dof, (len(keep), spec.n_dof)
            for k in ("dof_names", "dof_positions", "dof_velocities"):
                arrays[k] = np.ascontiguousarray(arrays[k][..., keep])
            tmp = dst + f".{os.getpid()}.tmp.npz"
            np.savez(tmp, **arrays)
            os.replace(tmp, dst)
        out.append(dst)
    return out


def algorithmic_bytes_per_env_step(spec: WorkloadSpec) -> int:
    """HBM bytes of a fused ideal, exactly SURVEY.md section 8d's accounting (2 682 B at K=2, 10 650 B at K=10 for
    G1; 2 310 B humanoid): the motion tables are cache-resident and excluded."""
    D, K, nd = spec.D, spec.K, spec.n_dof
    g1 = spec.robot == "g1"
    reads = D * 4 + (2 * nd * 4 + 8 if g1 else 0) + 8 + 16 + (K - 1) * D * 4
    policy = ((D - 12) + nd + 2) * 4 if g1 else D * 4
    writes = K * D * 4 + K * D * 4 + policy + 18
    return reads + writes


def disc_flops_per_row(in_dim: int, h1: int = 1024, h2: int = 512) -> int:
    return 2 * (in_dim * h1 + h1 * h2 + h2)


class HotPath:
    """All device state of one env shard + ``step()`` = one env-step of the hot path."""

    def __init__(self, spec: WorkloadSpec, num_envs: int, device, seed: int = 0, log_reward_terms: bool = False,
                 overlap: bool = False, fused_scaler: bool = True, disc_precision: str = "f16x3",
                 expert_stream: bool = False, fused_expert: bool = True, state: dict | None = None, state_sets: int = 1,
                 fused_tail: bool = True, one_call: bool = True):
        """``state``: use this synthetic state (a ``make_state`` dict on the device, e.g. a row block of a larger
        shard's state) instead of drawing one from ``seed``.  ``state_sets`` > 1: that many independently drawn input
        sets (seeds ``seed + 7919 i``), visited round-robin by successive steps, so that a benchmark's state reads are
        not served by the 256 MB Infinity Cache from the previous step's identical addresses.

        ``overlap``: run the discriminator on a second HIP stream so that the HBM-bound kernels of step t+1
        (motion sample, env step, compaction) execute under the MFMA-bound GEMMs of step t.  The style reward is
        consumed asynchronously in AMP (skrl reads it at the agent update), so nothing waits for it inside a step;
        ``synchronize()`` / ``torch.cuda.synchronize()`` joins both streams.  Bit-identical to the serial schedule
        (tests/test_gpu_disc.py).  Measured on MI355X it is a wash (+1.5 % at 65 536 envs, -10 % at 4 096): the GEMM
        workgroups already hold every wave slot, so the two streams time-slice instead of overlapping -> default off.

        ``fused_expert`` (default): the expert-motion sample shares the env step's launch (horizontal fusion:
        ``amp_env_step_with_reference``), so the two byte-moving kernels overlap instead of running back to back.

        ``expert_stream``: the expert-motion sample (``collect_reference``: the motion dataset's rows, no data
        dependence on the env state) runs on a side stream forked at the start of the step and joined at its end, i.e.
        under the env-step / compaction kernels instead of in front of them.  Measured: +1 % at 65 536 envs (both are
        HBM-bound and share the bandwidth), -22 % at 4 096 envs as a hipGraph (the cross-stream edges cost more than
        the 12 us kernel) -> default off."""
        self.spec, self.num_envs = spec, int(num_envs)
        self.overlap = bool(overlap)
        self.fused_tail = bool(fused_tail)  # compaction + finalize as one launch (amp_disc_style_reward_prescaled_compact)
        self.one_call = bool(one_call)      # the whole step as one amp_hot_step call on prebuilt arguments
        self.fused_scaler = bool(fused_scaler) and not self.overlap  # the overlapped schedule needs the snapshot pass
        self.device = nat.require_gpu(device)
        files = ",".join(clip_files(spec))
        self.motion = MotionLoader(files, self.device)
        ml = self.motion
        if spec.robot == "g1":
            perm, keys = ml.get_dof_index(spec.joint_names), ml.get_body_index(G1_KEY_BODY_NAMES)
        else:
            perm, keys = list(range(ml.num_dofs)), ml.get_body_index(HUMANOID_KEY_BODY_NAMES)
        D = ml.set_obs_layout(perm, ml.get_body_index([spec.reference_body])[0], keys)
        assert D == spec.D
        rewards = G1_REWARDS if spec.robot == "g1" else {}
        self.cfg = EnvStepConfig(n_dof=spec.n_dof, num_amp_observations=spec.K, max_episode_length=spec.max_episode_length,
                                 use_last_actions=spec.robot == "g1", reward_mode=1 if spec.robot == "g1" else 0, **rewards)
        self.kernel = EnvStepKernel(self.cfg, self.num_envs, self.device, log_reward_terms=log_reward_terms)
        if state is not None:
            if int(state["joint_pos"].shape[0]) != self.num_envs:
                raise nat.AmpEngineError("the injected state has the wrong number of envs")
            self.states = [state]
        else:
            self.states = [make_state(self.num_envs, spec.n_dof, spec.max_episode_length, ml.durations, seed + 7919 * i, self.device)
                           for i in range(max(1, int(state_sets)))]
        self.state = self.states[0]
        self.expert_obs = torch.zeros((self.num_envs, spec.K * D), device=self.device)
        # discriminator: torch.nn.Linear default init under torch.manual_seed(0); scaler mean 0 / var 1
        self.disc_weights = make_disc_weights(spec.K * D, seed=0)
        self.disc = AmpDiscriminator(self.disc_weights, self.device,
                                     running_mean=torch.zeros(spec.K * D, dtype=torch.float64),
                                     running_variance=torch.ones(spec.K * D, dtype=torch.float64),
                                     task_reward_weight=spec.task_weight, style_reward_weight=spec.style_weight,
                                     precision=disc_precision)
        # a shard whose whole batch takes the one-launch two-layer kernel: that kernel reads the AMP rows itself (scaler, clamp and
        # plane split on the 48 elements a lane holds), so the env step writes no discriminator input at all
        self.raw_rows = self.fused_scaler and self.fused_tail and self.one_call and bool(self.disc.plan_info(self.num_envs)["raw_input"])
        if self.raw_rows:
            self.fused_scaler = False
            self.kernel.amp_obs_read_next = True   # ... and keeps the rows in the cache hierarchy for it (default store policy)
        if self.fused_scaler:  # the env step emits the discriminator's scaled input directly (no separate scaler pass)
            self.kernel.attach_discriminator(self.disc)
        self._kernels = [self.kernel]
        # expert rows are a plausible AMP history to start from
        self.motion.collect_reference(self.state["motion_times"], self.state["motion_ids"], spec.K,
                                      out=self.kernel.amp_observation_buffer)
        sim_keys = ["joint_pos", "joint_vel", "joint_acc", "actions", "root_pos", "root_quat", "root_lin_vel", "root_ang_vel",
                    "body_pos", "soft_limits", "episode_length", "command", "last_actions"]
        if spec.robot != "g1":
            sim_keys = [k for k in sim_keys if k not in ("joint_acc", "actions", "soft_limits", "command", "last_actions")]
        self._sims = [{k: st[k] for k in sim_keys} for st in self.states]
        self._sim = self._sims[0]
        self.last = None
        self._n = 0
        self.fused_expert = bool(fused_expert) and not expert_stream
        self._hot_args = None
        self._expert_stream = None
        if expert_stream:
            self._expert_stream = torch.cuda.Stream(device=self.device)
            self._fork, self._join = torch.cuda.Event(), torch.cuda.Event()
        if self.overlap:
            self._disc_stream = torch.cuda.Stream(device=self.device)
            self._obs_ready = torch.cuda.Event()
            self._consumed = [torch.cuda.Event(), torch.cuda.Event()]
            for ev in self._consumed:  # torch creates the hipEvent lazily: force it, the C ABI needs the handle
                ev.record(self._disc_stream)
            # warm both workspaces + output tensors on the side stream
            with torch.cuda.stream(self._disc_stream):
                for slot in (0, 1):
                    self.disc._workspace(self.num_envs, slot)

    def step(self):
        graphs = getattr(self, "_graphs", None)
        if graphs:
            g, out = graphs[self._n % len(graphs)]
            g.replay()
            self._n += 1
            self.last = out
            return self.last
        return self._eager_step()

    def _build_hot_args(self):
        """One prebuilt ``AmpHotStepArgs`` per input set: the whole step becomes ONE call across the C ABI
        (``amp_hot_step``), with nothing marshalled per step -- on small shards the four separate calls of the generic
        path cost the host more (~90 us) than the step costs the GPU (~64 us at 8 192 envs)."""
        import ctypes as C

        N, K = self.num_envs, self.spec.K
        ws = self.disc._workspace(N)
        self._hot_keep = []  # ctypes structs referenced by pointer from the args
        args = []
        for st, sim in zip(self.states, self._sims):
          per_parity = []
          for par, k in enumerate(self._kernels):
            s = k.sim_state(key_body_indexes=[0, 1, 2, 3], **sim)
            b, c = k._buffers(), k.compact_args()
            k.check_reference(st["motion_times"], st["motion_ids"], self.expert_obs)
            a = nat.AmpHotStepArgs()
            a.cfg, a.state, a.bufs = C.addressof(k._c), C.addressof(s), C.addressof(b)
            a.num_envs = N
            a.motion = self.motion._handle.value if hasattr(self.motion._handle, "value") else self.motion._handle
            a.times, a.motion_ids = st["motion_times"].data_ptr(), st["motion_ids"].data_ptr()
            a.n_samples, a.K, a.expert_out = N, K, self.expert_obs.data_ptr()
            a.disc = self.disc._handle.value
            a.reward_scale, a.task_weight, a.style_weight = self.disc.reward_scale, self.disc.task_reward_weight, self.disc.style_reward_weight
            a.logits = None  # style / combined are set per step
            a.workspace, a.compact = ws.data_ptr(), C.addressof(c)
            self._hot_keep.append((s, b, c))
            per_parity.append(a)
          args.append(per_parity)
        self._hot_args = args
        self._hot_lib = nat.load()
        # the raw pointers baked into the arguments: a later disc._workspace() growth or attach_discriminator() reallocation
        # must not leave amp_hot_step writing through stale addresses (the tensors themselves are kept alive above)
        self._hot_keep.append((ws, [k.disc_input for k in self._kernels]))
        self._hot_ptrs = self._hot_pointer_key()

    def _hot_pointer_key(self):
        ws = (self.disc._ws or {}).get(0)
        return (None if ws is None else ws.data_ptr(),) + tuple(None if k.disc_input is None else k.disc_input.data_ptr()
                                                                 for k in self._kernels)

    def _eager_step(self, which: int | None = None):
        i = (self._n if which is None else which) % len(self.states)
        s, k = self.states[i], self.kernel
        self.state, self._sim = s, self._sims[i]
        if self.one_call and (self.fused_scaler or self.raw_rows) and self.fused_tail and self.fused_expert and not self.overlap:
            import ctypes as C

            if self._hot_args is None or self._hot_ptrs != self._hot_pointer_key():
                self._build_hot_args()
            a = self._hot_args[i][0]
            # fresh [N, 1] outputs per step (a caller may keep earlier steps' results), everything else is prebuilt
            style = torch.empty((self.num_envs, 1), dtype=torch.float32, device=self.device)
            combined = torch.empty((self.num_envs, 1), dtype=torch.float32, device=self.device)
            a.style, a.combined = style.data_ptr(), combined.data_ptr()
            with torch.cuda.device(self.device):
                nat.check(self._hot_lib.amp_hot_step(C.byref(a), nat.stream_ptr()), "amp_hot_step")
            self._n += 1
            self.last = {"style": style, "combined": combined}
            return self.last
        env_stream = torch.cuda.current_stream(self.device)
        if self.overlap and self._n > 0:
            # the previous discriminator call must have read amp_obs / reward before this step shifts / rewrites them
            env_stream.wait_event(self._consumed[(self._n - 1) & 1])
        reference = None
        if self.fused_expert:
            reference = (self.motion, s["motion_times"], s["motion_ids"], self.expert_obs)
        elif self._expert_stream is None:
            self.motion.collect_reference(s["motion_times"], s["motion_ids"], self.spec.K, out=self.expert_obs)
        else:
            self._fork.record(env_stream)
            with torch.cuda.stream(self._expert_stream):
                self._expert_stream.wait_event(self._fork)
                self.motion.collect_reference(s["motion_times"], s["motion_ids"], self.spec.K, out=self.expert_obs)
                self._join.record(self._expert_stream)
        k.launch(nat.AMP_PHASE_ALL, key_body_indexes=[0, 1, 2, 3], reference=reference, **self._sim)
        amp = k.amp_observation_buffer.view(self.num_envs, -1)
        tail = self.fused_scaler and self.fused_tail
        raw_tail = self.fused_tail and not self.fused_scaler and not self.overlap   # style_reward(compact=k): the same ride on raw rows
        if not tail and not raw_tail:
            k.compact_resets()
        if self.fused_scaler:
            # fused tail: the reset-id compaction rides on the finalize launch (one launch fewer; same results)
            self.last = self.disc.style_reward_prescaled(k.disc_input, k.reward, compact=k if tail else None)
        elif not self.overlap:
            self.last = self.disc.style_reward(amp, k.reward, compact=k if raw_tail else None)
        else:
            slot = self._n & 1
            self._obs_ready.record(env_stream)
            with torch.cuda.stream(self._disc_stream):
                self._disc_stream.wait_event(self._obs_ready)
                self.last = self.disc.style_reward(amp, k.reward, inputs_consumed=self._consumed[slot], workspace_slot=slot)
        if self._expert_stream is not None:
            env_stream.wait_event(self._join)  # the step is complete only with its expert rows
        self._n += 1
        return self.last

    def capture(self, warmup: int = 3):
        """Capture one env-step into a hipGraph (every engine launch is asynchronous on the caller's stream and
        allocation-free, so the whole step is capturable); later ``step()`` calls replay it.  Pays off when the
        shard is small enough for the step to be launch-bound (a few thousand envs)."""
        if self.overlap:
            raise nat.AmpEngineError("graph capture and the overlapped schedule are mutually exclusive")
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        graphs = []
        for i in range(len(self.states)):  # one graph per input set (the state pointers are baked into the nodes)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._eager_step(which=i)
            graphs.append((g, out))
        self._graphs = graphs
        return self

    def synchronize(self):
        torch.cuda.synchronize(self.device)


class AgentSide:
    """What an AMP agent does with the hot path's output between two rollouts, on the engine (skrl's AMP is
    third-party: parity unpinned; sizes agents/skrl_g1_walk_amp_cfg.yaml:44-58,64-66,91): a rollout store of the shard's AMP rows
    (``record()`` after every env step), the replay and motion-dataset rings, the discriminator trainer on the hot path's
    discriminator, and ``update()`` = one ``AmpDiscriminatorUpdate`` over the stored rollout (with ``group``: the multi-rank flow,
    ONE all-gather per update).  ``batch_size`` is the GLOBAL minibatch."""

    def __init__(self, hot: HotPath, *, rollouts: int = 16, batch_size: int = 4096, learning_epochs: int = 6, mini_batches: int = 2,
                 replay_rows: int = 1_000_000, motion_rows: int = 200_000, seed: int = 0, group=None, rank: int = 0):
        from .engine import AmpDiscriminatorTrainer, AmpDiscriminatorUpdate, AmpReplayBuffer

        self.hot, self.rollouts = hot, int(rollouts)
        N, C = hot.num_envs, hot.spec.K * hot.spec.D
        dev = hot.device
        self.store = torch.empty((self.rollouts, N, C), device=dev)
        self.store.copy_(hot.kernel.amp_observation_buffer.view(1, N, C).expand(self.rollouts, N, C))  # valid rows before the first rollout
        self.replay = AmpReplayBuffer(int(replay_rows), C, dev, seed=seed + 1)
        self.motion_dataset = AmpReplayBuffer(int(motion_rows), C, dev, seed=seed + 2)
        # the motion dataset: expert rows at uniformly drawn (clip, time), as the agent fills it from collect_reference_motions
        # (g1_amp_env.py:445-486); every rank draws its own
        ml = hot.motion
        gen = np.random.default_rng(seed + 17 + 7919 * rank)
        durations = np.asarray(ml.durations, dtype=np.float64).reshape(-1)
        left = int(motion_rows)
        while left > 0:
            n = min(left, 65536)
            ids = gen.integers(0, len(durations), n)
            times = gen.uniform(0.0, 1.0, n) * durations[ids]
            self.motion_dataset.add_samples(ml.collect_reference(torch.from_numpy(times), torch.from_numpy(ids), hot.spec.K))
            left -= n
        self.trainer = AmpDiscriminatorTrainer(hot.disc, batch_size=batch_size, defer_refresh=True)
        self.updater = AmpDiscriminatorUpdate(self.trainer, self.replay, self.motion_dataset, learning_epochs=learning_epochs,
                                              mini_batches=mini_batches, seed=seed + 3 + rank, group=group)
        self._slot = 0

    def record(self) -> None:
        """Store the step's AMP rows (what skrl's rollout memory keeps of ``extras["amp_obs"]``)."""
        self.store[self._slot % self.rollouts].copy_(self.hot.kernel.amp_observation_buffer.view(self.hot.num_envs, -1))
        self._slot += 1

    def update(self):
        return self.updater.update(self.store)


def make_disc_weights(in_dim: int, seed: int = 0, hidden=(1024, 512)):
    """torch.nn.Linear default init under torch.manual_seed(seed) -> [(W, b)] * 3 on the CPU."""
    torch.manual_seed(seed)
    dims = (in_dim,) + tuple(hidden) + (1,)
    layers = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(3)]
    return [(l.weight.detach().clone(), l.bias.detach().clone()) for l in layers]
