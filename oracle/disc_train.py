"""Oracle: one discriminator training step of skrl's AMP agent (TEST INFRASTRUCTURE ONLY).  PARITY UNPINNED.

skrl (>= 1.4.3) is absent; this restates ``AMP._update``'s discriminator part [recalled] with torch autograd on the CPU:
scaler update (train=True) -> BCE-with-logits on (policy U replay) vs motion -> logit regularisation on the output
layer -> gradient penalty w.r.t. the (scaled) motion states -> weight decay -> x discriminator_loss_scale -> Adam.
Hyper-parameters: agents/skrl_g1_walk_amp_cfg.yaml:87-95 (loss scale 5.0, logit reg 0.05, gradient penalty 5.0,
weight decay 1e-4, learning rate 5e-5).
"""

from __future__ import annotations

import torch

from . import disc as odisc


def scaler_update(mean: torch.Tensor, var: torch.Tensor, count: float, x: torch.Tensor):
    """RunningStandardScaler._parallel_variance with the batch statistics of ``x`` (fp64 buffers)."""
    in_mean = torch.mean(x.double(), dim=0)
    in_var = torch.var(x.double(), dim=0)
    n = x.shape[0]
    delta = in_mean - mean
    total = count + n
    m2 = var * count + in_var * n + delta ** 2 * count * n / total
    return mean + delta * n / total, m2 / total, total


def loss_and_grads(weights, policy, replay, motion, mean, var, *, loss_scale=5.0, logit_reg=0.05, grad_penalty=5.0,
                   weight_decay=1e-4, dtype=torch.float32):
    """Returns (loss dict, [gW1, gb1, gW2, gb2, gW3, gb3]) for raw AMP observations; the scaler is applied, not updated."""
    params = []
    for w, b in weights:
        params += [w.clone().to(dtype).requires_grad_(True), b.clone().to(dtype).requires_grad_(True)]
    ws = [(params[0], params[1]), (params[2], params[3]), (params[4], params[5])]

    def fwd(x):
        h = x
        for i, (w, b) in enumerate(ws):
            h = torch.nn.functional.linear(h, w, b)
            if i < 2:
                h = torch.relu(h)
        return h

    sc = (lambda x: odisc.scale_states(x, mean, var)) if mean is not None else (lambda x: x)
    sp = sc(policy).to(dtype)
    sr = sc(replay).to(dtype)
    sm = sc(motion).to(dtype).clone().requires_grad_(True)
    lp, lr_, lm = fwd(sp), fwd(sr), fwd(sm)
    cat = torch.cat([lp, lr_], dim=0)
    bce = torch.nn.BCEWithLogitsLoss()
    pred = 0.5 * (bce(cat, torch.zeros_like(cat)) + bce(lm, torch.ones_like(lm)))
    reg = logit_reg * torch.sum(torch.square(torch.flatten(params[4])))
    g = torch.autograd.grad(lm, sm, grad_outputs=torch.ones_like(lm), create_graph=True, retain_graph=True, only_inputs=True)[0]
    gp = grad_penalty * torch.sum(torch.square(g), dim=-1).mean()
    wd = weight_decay * torch.sum(torch.square(torch.cat([torch.flatten(params[i]) for i in (0, 2, 4)], dim=-1)))
    total = loss_scale * (pred + reg + gp + wd)
    grads = torch.autograd.grad(total, params)
    return dict(total=total.detach(), prediction=pred.detach(), grad_penalty=gp.detach(), logit_reg=reg.detach(),
                weight_decay=wd.detach()), [x.detach() for x in grads]


def adam_step(params, grads, m, v, t, lr=5e-5, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad), one step; returns new (params, m, v)."""
    out_p, out_m, out_v = [], [], []
    for p, g, mi, vi in zip(params, grads, m, v):
        mi = beta1 * mi + (1 - beta1) * g
        vi = beta2 * vi + (1 - beta2) * g * g
        mhat = mi / (1 - beta1 ** t)
        vhat = vi / (1 - beta2 ** t)
        out_p.append(p - lr * mhat / (vhat.sqrt() + eps))
        out_m.append(mi)
        out_v.append(vi)
    return out_p, out_m, out_v
