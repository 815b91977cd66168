"""Synthetic simulator state for benchmarks and property tests (no Isaac Sim in this stack).

Distributions follow SURVEY.md section 8d: joint_pos ~ U(-1, 1.3), joint_vel ~ N(0, 1.5), joint_acc ~ N(0, 30),
actions ~ N(0, 0.5); root pos = (N(0,1), N(0,1), U(0.35, 0.95)) (10-15 % below the 0.5 m termination height),
unit root quaternion (wxyz), root lin/ang vel ~ N(0,1), key bodies = root + N(0, 0.4); episode length ~ randint;
motion times ~ U(0,1) * duration, ids ~ randint(n_clips); soft limits +-0.9*pi/2; command ~ U(-1, 1).
All tensors are SoA ``[N, ...]`` float32 on the given device, generated with a seeded torch.Generator on the CPU
(so the CPU baseline sees bit-identical inputs) and copied once.
"""

from __future__ import annotations

import math

import numpy as np
import torch


def make_state(num_envs: int, n_dof: int, max_episode_length: int, durations: np.ndarray, seed: int, device,
               n_key: int = 4) -> dict:
    g = torch.Generator().manual_seed(int(seed))
    N = int(num_envs)
    rn = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    ru = lambda *s: torch.rand(*s, generator=g)   # noqa: E731
    root_pos = torch.cat([rn(N, 2), ru(N, 1) * 0.6 + 0.35], dim=1)
    quat = rn(N, 4)
    quat = quat / quat.norm(dim=1, keepdim=True)
    ids = torch.randint(0, len(durations), (N,), generator=g)
    times = ru(N).double() * torch.from_numpy(np.asarray(durations, dtype=np.float64))[ids]
    lim = torch.tensor([[-0.9 * math.pi / 2, 0.9 * math.pi / 2]] * n_dof, dtype=torch.float32)
    cpu = dict(
        joint_pos=ru(N, n_dof) * 2.3 - 1.0,
        joint_vel=rn(N, n_dof) * 1.5,
        joint_acc=rn(N, n_dof) * 30.0,
        actions=rn(N, n_dof) * 0.5,
        last_actions=rn(N, n_dof) * 0.5,
        root_pos=root_pos,
        root_quat=quat,
        root_lin_vel=rn(N, 3),
        root_ang_vel=rn(N, 3),
        body_pos=root_pos[:, None, :] + rn(N, n_key, 3) * 0.4,   # the key bodies, packed [N, n_key, 3]
        soft_limits=lim,
        episode_length=torch.randint(0, int(max_episode_length), (N,), generator=g),
        command=ru(N, 2) * 2.0 - 1.0,
        motion_times=times,
        motion_ids=ids,
    )
    cpu = {k: v.contiguous() for k, v in cpu.items()}
    dev = torch.device(device)
    return cpu if dev.type == "cpu" else {k: v.to(dev) for k, v in cpu.items()}
