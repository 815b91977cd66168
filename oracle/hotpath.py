"""Oracle: one env-step of the hot path (SURVEY.md 8d's unit of work) on the CPU (TEST INFRASTRUCTURE ONLY).

Composes ``oracle.motion`` / ``oracle.env`` / ``oracle.disc`` the way ``humanoid_amp_amd.workloads.HotPath.step`` composes
the engine's kernels -- expert-motion sample, done bits + reset ids, task reward, sim AMP obs + K-history shift + policy
obs, scaler + discriminator + style / combined reward -- on the same synthetic state, and reports the largest deviation
of every output.  Used by ``__graft_entry__.smoke()`` and ``tests/test_gpu_hotpath_oracle.py`` (the timed path meeting
the oracle in one hop).  Each piece cites the reference lines it restates in its own module.
"""

from __future__ import annotations

import torch

from . import disc as odisc
from . import env as oenv


def layout(mt, robot: str, g1_joint_names=None, g1_key_names=None, humanoid_key_names=None):
    """(dof permutation, reference body, key bodies) of the clip table for a robot."""
    if robot == "g1":
        return ([mt.dof_names.index(n) for n in g1_joint_names], mt.body_names.index("pelvis"),
                [mt.body_names.index(n) for n in g1_key_names])
    return list(range(len(mt.dof_names))), mt.body_names.index("torso"), [mt.body_names.index(n) for n in humanoid_key_names]


def step(mt, lay, spec, cfg: dict, st: dict, buf: torch.Tensor, weights, *, max_episode_length: int):
    """One oracle env-step.  ``st``: the synthetic state (CPU tensors, ``humanoid_amp_amd.synthetic.make_state`` keys);
    ``buf``: the AMP history [N, K, D] BEFORE the step (shifted in place).  Returns the expected outputs."""
    perm, ref, keys = lay
    N, K, D = buf.shape
    g1 = spec.robot == "g1"
    exp = dict(expert=oenv.collect_reference(mt, st["motion_times"].numpy(), st["motion_ids"].numpy(), K, perm, ref, keys))
    died, tout = oenv.dones(st["episode_length"], max_episode_length, st["root_pos"][:, 2], 0.5)
    exp.update(died=died, time_out=tout, reset_ids=oenv.reset_env_ids(died, tout))
    if g1:
        lim = st["soft_limits"].unsqueeze(0).expand(N, -1, -1)
        task, _ = oenv.g1_task_reward(cfg, st["root_lin_vel"], st["root_quat"], st["command"], died, st["actions"], st["joint_pos"],
                                      lim, st["joint_acc"], st["joint_vel"])
    else:
        task = torch.ones(N)
    obs = oenv.compute_obs(st["joint_pos"], st["joint_vel"], st["root_pos"], st["root_quat"], st["root_lin_vel"],
                           st["root_ang_vel"], st["body_pos"])
    amp = oenv.shift_history(buf, obs)
    pol = oenv.actor_observation(obs, st["last_actions"], st["command"], use_command=True) if g1 else obs
    ref_out = odisc.forward(weights, amp, torch.zeros(K * D, dtype=torch.float64), torch.ones(K * D, dtype=torch.float64),
                            task=task.unsqueeze(-1), task_w=spec.task_weight, style_w=spec.style_weight)
    exp.update(task=task, amp=amp.clone(), policy=pol, style=ref_out["style"], combined=ref_out["combined"], logits=ref_out["logits"])
    return exp


def compare(hot, out: dict, exp: dict) -> dict:
    """Largest |engine - oracle| per output of one ``HotPath.step()`` (``hot.kernel`` holds the step's env outputs);
    done bits and reset ids are compared exactly (``*_equal`` entries)."""
    k, N = hot.kernel, hot.num_envs
    n = int(k.reset_count.item())
    mx = lambda a, b: float((a.cpu().reshape(-1) - b.reshape(-1)).abs().max())  # noqa: E731
    return dict(expert=mx(hot.expert_obs, exp["expert"]), amp=mx(k.amp_observation_buffer, exp["amp"]),
                policy=mx(k.policy_obs, exp["policy"]), task=mx(k.reward, exp["task"]), style=mx(out["style"], exp["style"]),
                combined=mx(out["combined"], exp["combined"]),
                dones_equal=bool(torch.equal(k.died.cpu(), exp["died"]) and torch.equal(k.time_out.cpu(), exp["time_out"])),
                reset_ids_equal=bool(n == exp["reset_ids"].numel() and torch.equal(k.reset_ids[:n].cpu(), exp["reset_ids"])),
                n_reset=n)
