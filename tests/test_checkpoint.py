"""skrl-checkpoint loader (host logic on CPU; the GPU leg builds the discriminator from the file)."""

import collections

import pytest
import torch

from oracle import disc as odisc


def _fake_agent_checkpoint(in_dim=166, seed=0):
    """Same structure skrl's AMP agent writes (key names per its generated model source) [recalled]."""
    w = odisc.make_weights(in_dim, seed=seed)
    disc = collections.OrderedDict([
        ("net_container.0.weight", w[0][0]), ("net_container.0.bias", w[0][1]),
        ("net_container.2.weight", w[1][0]), ("net_container.2.bias", w[1][1]),
        ("output_layer.weight", w[2][0]), ("output_layer.bias", w[2][1]),
    ])
    g = torch.Generator().manual_seed(seed + 1)
    scaler = {"running_mean": torch.randn(in_dim, generator=g, dtype=torch.float64) * 0.2,
              "running_variance": torch.rand(in_dim, generator=g, dtype=torch.float64) + 0.3,
              "current_count": torch.tensor(12345.0, dtype=torch.float64)}
    policy = collections.OrderedDict([("log_std_parameter", torch.zeros(29)), ("net_container.0.weight", torch.zeros(1024, 102)),
                                      ("net_container.0.bias", torch.zeros(1024))])
    return {"policy": policy, "value": {}, "discriminator": disc, "amp_state_preprocessor": scaler,
            "state_preprocessor": {}, "value_preprocessor": {}}, w, scaler


def test_parse_and_validate(tmp_path):
    from humanoid_amp_amd.checkpoint import load_skrl_checkpoint, parse_skrl_checkpoint

    ck, w, scaler = _fake_agent_checkpoint()
    layers, mean, var = parse_skrl_checkpoint(ck, amp_observation_size=166)
    for (a, b), (c, d) in zip(layers, w):
        assert torch.equal(a, c) and torch.equal(b, d)
    assert torch.equal(mean, scaler["running_mean"]) and var.dtype == torch.float64
    path = tmp_path / "agent_1000.pt"
    torch.save(ck, path)
    layers2, mean2, _ = load_skrl_checkpoint(str(path))   # weights_only=True load
    assert torch.equal(layers2[2][0], w[2][0]) and torch.equal(mean2, mean)
    # a bare discriminator state dict works too, without a scaler
    layers3, m3, v3 = parse_skrl_checkpoint(ck["discriminator"])
    assert len(layers3) == 3 and m3 is None and v3 is None
    with pytest.raises(ValueError, match="K \\* D"):
        parse_skrl_checkpoint(ck, amp_observation_size=830)
    bad = dict(ck, discriminator={k: v for k, v in ck["discriminator"].items() if not k.startswith("output_layer")})
    with pytest.raises(ValueError, match="3 Linear layers"):
        parse_skrl_checkpoint(bad)
    with pytest.raises(ValueError):
        parse_skrl_checkpoint([1, 2, 3])


@pytest.mark.gpu
def test_discriminator_from_checkpoint_matches_oracle(tmp_path):
    from humanoid_amp_amd.checkpoint import discriminator_from_checkpoint

    ck, w, scaler = _fake_agent_checkpoint(seed=4)
    path = tmp_path / "agent.pt"
    torch.save(ck, path)
    d = discriminator_from_checkpoint(str(path), "cuda:0", amp_observation_size=166, discriminator_reward_scale=2.0,
                                      task_reward_weight=0.5, style_reward_weight=0.5)
    x = torch.randn(500, 166, generator=torch.Generator().manual_seed(2))
    task = torch.randn(500, 1, generator=torch.Generator().manual_seed(3))
    out = d.style_reward(x.cuda(), task.cuda())
    ref = odisc.forward(w, x, scaler["running_mean"], scaler["running_variance"], task=task, task_w=0.5, style_w=0.5)
    assert float((out["style"].cpu() - ref["style"]).abs().max()) <= 1e-5
    assert float((out["combined"].cpu() - ref["combined"]).abs().max()) <= 1e-5
