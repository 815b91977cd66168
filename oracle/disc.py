"""Oracle: AMP discriminator style reward (TEST INFRASTRUCTURE ONLY).  PARITY UNPINNED.

The discriminator, its input scaler and the style-reward formula live in skrl's ``AMP`` agent
(``skrl >= 1.4.3``, train.py:123-129 of the reference), which is neither vendored in /root/reference nor
installed here.  This file restates the published algorithm; the reference pins only the
shape ``[K*D -> 1024 -> 512 -> 1]`` with ReLU (agents/skrl_g1_walk_amp_cfg.yaml:31-39), the scaler class
(:77-78) and the scales (:88-95).
"""

from __future__ import annotations

import torch


def make_weights(in_dim: int, seed: int = 0, hidden=(1024, 512)):
    """``torch.nn.Linear`` default init under ``torch.manual_seed(seed)``; returns [(W [out,in], b [out])]*3."""
    torch.manual_seed(seed)
    dims = (in_dim,) + tuple(hidden) + (1,)
    layers = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
    return [(l.weight.detach().clone(), l.bias.detach().clone()) for l in layers]


def scale_states(x: torch.Tensor, running_mean: torch.Tensor, running_variance: torch.Tensor,
                 epsilon: float = 1e-8, clip: float = 5.0) -> torch.Tensor:
    """skrl RunningStandardScaler forward (stats kept in fp64, cast to fp32 for the arithmetic)."""
    return torch.clamp((x - running_mean.float()) / (torch.sqrt(running_variance.float()) + epsilon), min=-clip, max=clip)


def logits(weights, xs: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    h = xs.to(dtype)
    n = len(weights)
    for i, (w, b) in enumerate(weights):
        h = torch.nn.functional.linear(h, w.to(dtype), b.to(dtype))
        if i < n - 1:
            h = torch.relu(h)
    return h


def style_reward(lg: torch.Tensor, scale: float = 2.0) -> torch.Tensor:
    """-log(max(1 - sigmoid(logit), 1e-4)) * discriminator_reward_scale."""
    r = -torch.log(torch.maximum(1 - 1 / (1 + torch.exp(-lg)), torch.tensor(0.0001, dtype=lg.dtype)))
    return r * scale


def combine(task: torch.Tensor, style: torch.Tensor, task_w: float, style_w: float) -> torch.Tensor:
    return task_w * task + style_w * style


def forward(weights, amp_obs, running_mean=None, running_variance=None, *, reward_scale=2.0, task=None,
            task_w=0.0, style_w=1.0):
    with torch.no_grad():
        xs = amp_obs if running_mean is None else scale_states(amp_obs, running_mean, running_variance)
        lg = logits(weights, xs)
        st = style_reward(lg, reward_scale)
        out = dict(scaled=xs, logits=lg, style=st)
        if task is not None:
            out["combined"] = combine(task, st, task_w, style_w)
    return out
