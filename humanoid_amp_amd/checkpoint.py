"""skrl checkpoint -> discriminator weights + AMP state scaler.

The reference trains with skrl's ``AMP`` agent and resumes / evaluates from ``checkpoints/agent_<step>.pt``
(train.py:244-247,292-295; play.py:205-208; play_deploy.py:40-63).  skrl writes one dict per agent:
``{"policy": sd, "value": sd, "discriminator": sd, "optimizer": ..., "state_preprocessor": sd,
"value_preprocessor": sd, "amp_state_preprocessor": sd}`` where the discriminator built from
``agents/skrl_g1_walk_amp_cfg.yaml:31-39`` is ``net_container = Sequential(Linear, ReLU, Linear, ReLU)`` +
``output_layer = Linear(512, 1)`` and a RunningStandardScaler state dict holds ``running_mean``,
``running_variance`` (float64) and ``current_count``  [recalled: skrl >= 1.4 is absent here, so the key names are
matched structurally -- every 2-D ``*.weight`` with its ``*.bias``, in order -- rather than literally].

Files are opened with ``torch.load(..., weights_only=True)`` only: nothing in the file is executed.
"""

from __future__ import annotations

from typing import Mapping

import torch


def _linear_layers(state_dict: Mapping[str, torch.Tensor]):
    layers = []
    for name, w in state_dict.items():
        if name.endswith("weight") and isinstance(w, torch.Tensor) and w.dim() == 2:
            bias_name = name[: -len("weight")] + "bias"
            if bias_name not in state_dict:
                raise ValueError(f"checkpoint layer {name} has no bias")
            layers.append((w.detach().float(), state_dict[bias_name].detach().float()))
    return layers


def parse_skrl_checkpoint(obj, amp_observation_size: int | None = None):
    """(weights, running_mean, running_variance) from a loaded skrl agent checkpoint (or a bare discriminator
    state dict).  ``weights`` = [(W1, b1), (W2, b2), (W3, b3)] in torch.nn.Linear layout; the scaler entries are
    ``None`` when the checkpoint carries no ``amp_state_preprocessor``."""
    if not isinstance(obj, Mapping):
        raise ValueError("a skrl checkpoint is a dict of state dicts")
    disc = obj.get("discriminator", obj)
    if not isinstance(disc, Mapping):
        raise ValueError("checkpoint['discriminator'] is not a state dict")
    layers = _linear_layers(disc)
    if len(layers) != 3:
        raise ValueError(f"expected 3 Linear layers in the discriminator, found {len(layers)}")
    (w1, b1), (w2, b2), (w3, b3) = layers
    if w2.shape[1] != w1.shape[0] or w3.shape != (1, w2.shape[0]) or b1.numel() != w1.shape[0] or b2.numel() != w2.shape[0] \
            or b3.numel() != 1:
        raise ValueError("discriminator layer shapes do not chain: "
                         f"{tuple(w1.shape)} -> {tuple(w2.shape)} -> {tuple(w3.shape)}")
    if amp_observation_size is not None and w1.shape[1] != amp_observation_size:
        raise ValueError(f"discriminator expects {w1.shape[1]} inputs, the env emits {amp_observation_size} (K * D)")
    mean = var = None
    scaler = obj.get("amp_state_preprocessor")
    if isinstance(scaler, Mapping) and "running_mean" in scaler:
        mean = scaler["running_mean"].detach().double().reshape(-1)
        var = scaler["running_variance"].detach().double().reshape(-1)
        if mean.numel() != w1.shape[1] or var.numel() != w1.shape[1]:
            raise ValueError("amp_state_preprocessor statistics do not match the discriminator input size")
    return layers, mean, var


def load_skrl_checkpoint(path: str, amp_observation_size: int | None = None):
    """``parse_skrl_checkpoint`` of a ``.pt`` file, loaded without unpickling arbitrary objects."""
    obj = torch.load(path, map_location="cpu", weights_only=True)
    return parse_skrl_checkpoint(obj, amp_observation_size)


def discriminator_from_checkpoint(path: str, device, amp_observation_size: int | None = None, **kwargs):
    """An :class:`~humanoid_amp_amd.engine.AmpDiscriminator` initialised from a skrl agent checkpoint; ``kwargs`` are
    the reward scales of the agent YAML (agents/*.yaml:88-95)."""
    from .engine import AmpDiscriminator

    weights, mean, var = load_skrl_checkpoint(path, amp_observation_size)
    return AmpDiscriminator(weights, device, running_mean=mean, running_variance=var, **kwargs)
