"""GPU: the discriminator training step vs the torch-autograd oracle.  PARITY UNPINNED (skrl absent): the oracle
restates skrl's AMP._update discriminator part (oracle/disc_train.py).

Bars: loss terms <= 1e-5 relative; every gradient <= 2e-5 * max|grad| of its tensor (fp32 GEMM reductions over
12 288 rows are ordered differently); Adam-updated weights <= 1e-6 absolute after one step; scaler statistics
<= 1e-9 relative (fp64)."""

import numpy as np
import pytest
import torch

from oracle import disc as odisc
from oracle import disc_train as odt

pytestmark = pytest.mark.gpu


def _batches(in_dim, B, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, in_dim, generator=g) * s + o for s, o in ((1.0, 0.0), (1.2, 0.1), (0.8, -0.2))]


def _split(flat, weights):
    out, i = [], 0
    for w, b in weights:
        for t in (w, b):
            out.append(flat[i:i + t.numel()].view_as(t))
            i += t.numel()
    return out


@pytest.mark.parametrize("gemm_precision", ["f16x3", "f32"])  # fp16-split GEMMs (default) / everything on the fp32 pipe
@pytest.mark.parametrize("in_dim,B", [(166, 512), (830, 256), (162, 1000)])
def test_gradients_and_loss_no_scaler(in_dim, B, gemm_precision):
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer

    w = odisc.make_weights(in_dim, seed=in_dim)
    p, r, m = _batches(in_dim, B, seed=B)
    disc = AmpDiscriminator([(a.cuda(), b.cuda()) for a, b in w], "cuda:0")
    tr = AmpDiscriminatorTrainer(disc, batch_size=B, use_scaler=False, update_scaler=False, apply_update=False,
                                 gemm_precision=gemm_precision)
    out = tr.step(p.cuda(), r.cuda(), m.cuda(), want_grads=True)
    L, G = odt.loss_and_grads(w, p, r, m, None, None)
    L64, G64 = odt.loss_and_grads(w, p, r, m, None, None, dtype=torch.float64)
    for name, key in (("prediction", "prediction"), ("grad_penalty", "grad_penalty"), ("logit_reg", "logit_reg"),
                      ("weight_decay", "weight_decay")):
        assert abs(float(out[name]) - float(L64[key])) <= 1e-5 * max(1.0, abs(float(L64[key]))), name
    assert abs(float(out["loss"]) - float(L64["total"])) <= 2e-5 * abs(float(L64["total"]))
    got = _split(out["grads"].cpu(), w)
    for i, (g_gpu, g_ref, g_64) in enumerate(zip(got, G, G64)):
        scale = float(g_64.abs().max())
        err_gpu = float((g_gpu.double() - g_64).abs().max())
        err_cpu = float((g_ref.double() - g_64).abs().max())
        assert err_gpu <= err_cpu + 2e-5 * scale, (i, err_gpu, err_cpu, scale)
    # apply_update=False: the weights did not move
    for (a, b), (c, d) in zip(tr.weights(), w):
        assert torch.equal(a.cpu(), c) and torch.equal(b.cpu(), d)


@pytest.mark.parametrize("gemm_precision", ["f16x3", "f32"])
def test_full_step_with_scaler_and_adam(gemm_precision):
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer

    in_dim, B = 166, 768
    w = odisc.make_weights(in_dim, seed=1)
    g = torch.Generator().manual_seed(5)
    mean0 = torch.randn(in_dim, generator=g, dtype=torch.float64) * 0.1
    var0 = torch.rand(in_dim, generator=g, dtype=torch.float64) + 0.5
    count0 = 1000.0
    disc = AmpDiscriminator([(a.cuda(), b.cuda()) for a, b in w], "cuda:0", running_mean=mean0, running_variance=var0)
    tr = AmpDiscriminatorTrainer(disc, batch_size=B, running_mean=mean0, running_variance=var0, current_count=count0,
                                 learning_rate=1e-3, gemm_precision=gemm_precision)
    # oracle state
    params = [t.clone() for wb in w for t in wb]
    mo = [torch.zeros_like(t) for t in params]
    ve = [torch.zeros_like(t) for t in params]
    mean, var, count = mean0.clone(), var0.clone(), count0
    for step in range(1, 4):
        p, r, m = _batches(in_dim, B, seed=100 + step)
        out = tr.step(p.cuda(), r.cuda(), m.cuda(), want_grads=True)
        # skrl order: each batch updates the running statistics, then is scaled with them
        scaled = []
        for x in (p, r, m):
            mean, var, count = odt.scaler_update(mean, var, count, x)
            scaled.append(odisc.scale_states(x, mean, var))
        cur = [(params[0], params[1]), (params[2], params[3]), (params[4], params[5])]
        L, G = odt.loss_and_grads(cur, *scaled, None, None)
        assert abs(float(out["loss"]) - float(L["total"])) <= 5e-5 * abs(float(L["total"])), step
        g_gpu = _split(out["grads"].cpu(), w)
        for a, b in zip(g_gpu, G):
            assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max())
        # Adam's m / (sqrt(v) + eps) is sign-like for near-zero gradients, so feeding each side its own (1e-5-close)
        # gradients would compare noise; the Adam arithmetic is checked on the engine's gradients instead
        params, mo, ve = odt.adam_step(params, g_gpu, mo, ve, step, lr=1e-3)
    gm, gv, gc = tr.scaler_state()
    assert gc == count
    assert float(((gm.cpu() - mean) / (mean.abs() + 1e-6)).abs().max()) <= 1e-9
    assert float(((gv.cpu() - var) / var).abs().max()) <= 1e-9
    for (a, b), (c, d) in zip(tr.weights(), [(params[0], params[1]), (params[2], params[3]), (params[4], params[5])]):
        assert float((a.cpu() - c).abs().max()) <= 1e-6 and float((b.cpu() - d).abs().max()) <= 1e-6
    # the discriminator handle serves inference with the trained weights and the updated scaler
    x = _batches(in_dim, 300, seed=9)[0]
    ref = odisc.forward([(params[0], params[1]), (params[2], params[3]), (params[4], params[5])], x, mean, var)
    got = disc.style_reward(x.cuda(), want_logits=True)
    assert float((got["logits"].cpu() - ref["logits"]).abs().max()) <= 5e-5


@pytest.mark.parametrize("in_dim", [166, 830])
def test_f16x3_gemms_match_the_fp32_pipe_at_full_size(in_dim):
    """BASELINE's minibatch (3 x 4096 rows): the step whose large GEMMs run on the fp16 matrix pipe (two planes per operand,
    three MFMAs per product, split-K weight gradients) against the same step on the fp32 pipe -- loss terms and every
    gradient tensor agree to 1e-5 of the tensor's scale (both are ~1e-6 from an fp64 evaluation at the small sizes above)."""
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer

    B = 4096
    w = odisc.make_weights(in_dim, seed=3)
    p, r, m = _batches(in_dim, B, seed=77)
    outs = {}
    for prec in ("f16x3", "f32"):
        disc = AmpDiscriminator([(a.cuda(), b.cuda()) for a, b in w], "cuda:0")
        tr = AmpDiscriminatorTrainer(disc, batch_size=B, use_scaler=False, update_scaler=False, apply_update=False,
                                     gemm_precision=prec)
        outs[prec] = tr.step(p.cuda(), r.cuda(), m.cuda(), want_grads=True)
    for name in ("prediction", "grad_penalty", "logit_reg", "weight_decay", "loss"):
        a, b = float(outs["f16x3"][name]), float(outs["f32"][name])
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (name, a, b)
    ga, gb = _split(outs["f16x3"]["grads"].cpu(), w), _split(outs["f32"]["grads"].cpu(), w)
    for i, (a, b) in enumerate(zip(ga, gb)):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), i


def test_captured_training_step_equals_eager():
    """AmpDiscriminatorTrainer.capture(): the step as ONE hipGraph (scaler count, Adam step and bias corrections live on the
    device and are advanced by the graph itself) replayed five times == five eager steps of an identical trainer, bit for
    bit: weights, running statistics, sample count, losses."""
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer
    from humanoid_amp_amd.workloads import make_disc_weights

    B, dim = 512, 166
    gen = torch.Generator().manual_seed(4)
    batches = [[(torch.randn(B, dim, generator=gen) * (1.0 + 0.3 * k)).cuda() for k in range(3)] for _ in range(7)]

    def make():
        disc = AmpDiscriminator(make_disc_weights(dim, seed=1), "cuda:0")
        return disc, AmpDiscriminatorTrainer(disc, batch_size=B, learning_rate=1e-3)

    (d_e, t_e), (d_g, t_g) = make(), make()
    for b in batches[:2]:
        t_e.step(*b)
        t_g.step(*b)
    t_g.capture()
    for b in batches[2:]:
        le = t_e.step(*b)
        lg = t_g.step_captured(*b)
        for k in AmpDiscriminatorTrainer.LOSS_TERMS:
            assert torch.equal(le[k], lg[k]), k
    for (we, be), (wg, bg) in zip(t_e.weights(), t_g.weights()):
        assert torch.equal(we, wg) and torch.equal(be, bg)
    me, ve, ce = t_e.scaler_state()
    mg, vg, cg = t_g.scaler_state()
    assert torch.equal(me, mg) and torch.equal(ve, vg) and ce == cg == 1.0 + 7 * 3 * B
    x = batches[0][0]
    assert torch.equal(d_e.style_reward(x)["style"], d_g.style_reward(x)["style"])  # the inference planes followed


def test_deferred_refresh_gives_the_same_discriminator():
    """defer_refresh=True: the steps leave the discriminator's inference-side derived data stale; after refresh() the handle
    scores exactly like one trained with the default per-step refresh (same weights, same scaler, same planes)."""
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer
    from humanoid_amp_amd.workloads import make_disc_weights

    B, dim = 512, 166
    gen = torch.Generator().manual_seed(6)
    batches = [[(torch.randn(B, dim, generator=gen) * (1.0 + 0.2 * k)).cuda() for k in range(3)] for _ in range(4)]
    x = (torch.randn(3000, dim, generator=gen) * 1.3).cuda()   # the LDS-DMA-free register-staged plan; planes are used either way
    outs = []
    for defer in (False, True):
        disc = AmpDiscriminator(make_disc_weights(dim, seed=2), "cuda:0")
        tr = AmpDiscriminatorTrainer(disc, batch_size=B, learning_rate=2e-3, defer_refresh=defer)
        before = disc.style_reward(x)["style"].clone()
        for b in batches:
            tr.step(*b)
        if defer:
            # until refresh() the handle is in between (trained biases / output layer, weight planes and scaler vectors from
            # before the training): its scores are undefined and must not be used -- only finite
            assert bool(torch.isfinite(disc.style_reward(x)["style"]).all())
            tr.refresh()
        outs.append(disc.style_reward(x, want_logits=True))
    assert torch.equal(outs[0]["logits"], outs[1]["logits"]) and torch.equal(outs[0]["style"], outs[1]["style"])
    assert not torch.equal(outs[0]["style"], before)


@pytest.mark.parametrize("in_dim,B", [(166, 4096), (830, 1024)])
def test_forked_penalty_chain_equals_the_in_line_step(in_dim, B, monkeypatch):
    """The gradient-penalty chain runs on the trainer's side stream beside the prediction loss's backward (fork / join through
    events).  Against the same step kept on one stream (AMP_TRAIN_FORK=0): loss terms equal, every gradient within 2e-6 of its
    tensor's scale (the two weight-gradient products of a weight are summed slice after slice instead of pairwise); and the
    forked step is deterministic -- two identical trainers stay bit-identical over five updating steps (a race between the two
    streams would show here)."""
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer

    w = odisc.make_weights(in_dim, seed=5)
    p, r, m = (t.cuda() for t in _batches(in_dim, B, seed=31))
    outs = {}
    for fork in ("1", "0"):
        monkeypatch.setenv("AMP_TRAIN_FORK", fork)
        disc = AmpDiscriminator([(a.cuda(), b.cuda()) for a, b in w], "cuda:0")
        tr = AmpDiscriminatorTrainer(disc, batch_size=B, use_scaler=False, update_scaler=False, apply_update=False)
        outs[fork] = tr.step(p, r, m, want_grads=True)
        torch.cuda.synchronize()
    for name in AmpDiscriminatorTrainer.LOSS_TERMS:
        a, b = float(outs["1"][name]), float(outs["0"][name])
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (name, a, b)
    for i, (a, b) in enumerate(zip(_split(outs["1"]["grads"].cpu(), w), _split(outs["0"]["grads"].cpu(), w))):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), i

    monkeypatch.setenv("AMP_TRAIN_FORK", "1")
    gen = torch.Generator().manual_seed(8)
    batches = [[(torch.randn(B, in_dim, generator=gen) * (1.0 + 0.3 * k)).cuda() for k in range(3)] for _ in range(5)]
    trained = []
    for _ in range(2):
        disc = AmpDiscriminator([(a.cuda(), b.cuda()) for a, b in w], "cuda:0")
        tr = AmpDiscriminatorTrainer(disc, batch_size=B, learning_rate=1e-3)
        losses = [tr.step(*b) for b in batches]
        trained.append((tr.weights(), [{k: v.clone() for k, v in l.items() if k in AmpDiscriminatorTrainer.LOSS_TERMS} for l in losses]))
    for (wa, ba), (wb, bb) in zip(trained[0][0], trained[1][0]):
        assert torch.equal(wa, wb) and torch.equal(ba, bb)
    for la, lb in zip(trained[0][1], trained[1][1]):
        for k in la:
            assert torch.equal(la[k], lb[k]), k
