"""GPU parity: discriminator style reward (fp32 MFMA) vs the oracle.  PARITY UNPINNED by the reference (skrl absent).

Bar: |gpu - f64| <= |cpu32 - f64| + 1e-5 on logits (fp64 arbitration) and <= 1e-5 absolute on
the style / combined reward vs the fp32 oracle.
"""

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import disc as odisc

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _check(weights, x, mean, var, task, task_w, style_w, logit_scale=1.0):
    from humanoid_amp_amd.engine import AmpDiscriminator

    disc = AmpDiscriminator([(w.cuda(), b.cuda()) for w, b in weights], "cuda:0", running_mean=mean, running_variance=var,
                            task_reward_weight=task_w, style_reward_weight=style_w)
    out = disc.style_reward(x.cuda(), task.cuda() if task is not None else None, want_logits=True)
    ref = odisc.forward(weights, x, mean, var, task=task, task_w=task_w, style_w=style_w)
    with torch.no_grad():
        xs = ref["scaled"]
        lg64 = odisc.logits(weights, xs, dtype=torch.float64)
    g = out["logits"].cpu().double()
    err_gpu = (g - lg64).abs()
    err_cpu = (ref["logits"].double() - lg64).abs()
    assert float((err_gpu - err_cpu).max()) <= TOL * logit_scale, (float(err_gpu.max()), float(err_cpu.max()))
    assert float(err_gpu.max()) <= 2e-5 * logit_scale
    # d(style)/d(logit) <= reward_scale = 2: the style bar follows the logit bar
    assert float((out["style"].cpu() - ref["style"]).abs().max()) <= TOL * logit_scale
    if task is not None:
        assert float((out["combined"].cpu() - ref["combined"]).abs().max()) <= TOL * logit_scale
    return out


def test_disc_golden_fixture():
    fx = gu.golden("disc_k2_166")
    w = odisc.make_weights(166, seed=int(fx["seed"]))
    out = _check(w, torch.from_numpy(fx["amp_obs"]), torch.from_numpy(fx["running_mean"]), torch.from_numpy(fx["running_variance"]),
                 torch.from_numpy(fx["task_reward"]), 0.5, 0.5)
    assert np.max(np.abs(out["logits"].cpu().numpy() - fx["logits_f64"])) <= 2e-5
    assert np.max(np.abs(out["style"].cpu().numpy() - fx["style_reward"])) <= TOL


@pytest.mark.parametrize("in_dim,rows", [(166, 1), (166, 127), (166, 128), (166, 129), (162, 1000), (830, 513), (166, 4096)])
def test_disc_shapes_and_ragged_rows(in_dim, rows):
    g = torch.Generator().manual_seed(in_dim + rows)
    w = odisc.make_weights(in_dim, seed=rows)
    x = torch.randn(rows, in_dim, generator=g) * 2.0
    mean = torch.randn(in_dim, generator=g, dtype=torch.float64) * 0.2
    var = torch.rand(in_dim, generator=g, dtype=torch.float64) + 0.1
    task = torch.randn(rows, 1, generator=g)
    _check(w, x, mean, var, task, 0.3, 0.7)


def test_disc_no_scaler_and_saturation():
    """Without a scaler the input is used as is; huge logits hit the 1e-4 floor of the style reward."""
    w = odisc.make_weights(166, seed=11)
    w[2] = (w[2][0] * 400.0, w[2][1])
    x = torch.randn(300, 166, generator=torch.Generator().manual_seed(1))
    out = _check(w, x, None, None, None, 0.0, 1.0, logit_scale=400.0)  # logits are O(100) here: relative bar
    assert float(out["style"].max()) <= -np.log(1e-4) * 2.0 + 1e-4
    assert float(out["style"].min()) >= 0.0


def test_disc_full_size_linearity_property():
    """65 536 rows (BASELINE size): the pre-activation of layer 3 is linear in w3, so doubling (w3, b3) must
    double every logit exactly (power-of-two scaling is exact in fp32)."""
    from humanoid_amp_amd.engine import AmpDiscriminator

    w = odisc.make_weights(166, seed=5)
    x = torch.randn(65536, 166, generator=torch.Generator().manual_seed(2)).cuda()
    a = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0").style_reward(x, want_logits=True)["logits"]
    w2 = [w[0], w[1], (w[2][0] * 2.0, w[2][1] * 2.0)]
    b = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w2], "cuda:0").style_reward(x, want_logits=True)["logits"]
    assert torch.equal(a * 2.0, b)
    # and a row's logit does not depend on which tile / position it sits in
    perm = torch.randperm(65536, generator=torch.Generator().manual_seed(3)).cuda()
    c = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0").style_reward(x[perm].contiguous(), want_logits=True)["logits"]
    assert torch.equal(c, a[perm])
    sub = odisc.forward(w, x[:256].cpu())
    assert float((a[:256].cpu() - sub["logits"]).abs().max()) <= 2e-5


def test_overlapped_hot_path_matches_serial():
    """Two-stream pipelining (discriminator of step t under the env kernels of step t+1) must not change a bit:
    same style / combined rewards, same buffers as the serial schedule, over several steps."""
    import contextlib
    import io

    from humanoid_amp_amd.workloads import WORKLOADS, HotPath

    outs = {}
    for overlap in (False, True):
        with contextlib.redirect_stdout(io.StringIO()):
            hot = HotPath(WORKLOADS["g1_walk"], 20000, "cuda:0", seed=3, overlap=overlap)
        rec = []
        for _ in range(6):
            r = hot.step()
            rec.append((r["style"], r["combined"]))
        hot.synchronize()
        outs[overlap] = (rec, hot.kernel.amp_observation_buffer.clone(), hot.kernel.reward.clone(), hot.kernel.reset_ids.clone())
    for (s0, c0), (s1, c1) in zip(outs[False][0], outs[True][0]):
        assert torch.equal(s0, s1) and torch.equal(c0, c1)
    for a, b in zip(outs[False][1:], outs[True][1:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("workload,envs", [("g1_walk", 5000), ("g1_dance", 700), ("humanoid3", 3000)])
def test_fused_scaler_matches_separate_pass(workload, envs):
    """amp_env_step's fused discriminator input (scaled + padded) == the stand-alone scaler pass, bit for bit, so the
    style / combined rewards are identical; checked over steps so every history slot has been shifted through."""
    import contextlib
    import io

    from humanoid_amp_amd.workloads import WORKLOADS, HotPath

    res = {}
    for fused in (False, True):
        with contextlib.redirect_stdout(io.StringIO()):
            hot = HotPath(WORKLOADS[workload], envs, "cuda:0", seed=5, fused_scaler=fused)
        # a non-trivial scaler
        g = torch.Generator().manual_seed(9)
        kd = hot.spec.K * hot.spec.D
        hot.disc.set_scaler(torch.randn(kd, generator=g, dtype=torch.float64) * 0.3, torch.rand(kd, generator=g, dtype=torch.float64) + 0.2)
        if fused:
            hot.kernel.attach_discriminator(hot.disc)
        outs = [hot.step() for _ in range(hot.spec.K + 1)]
        hot.synchronize()
        res[fused] = (outs[-1]["style"].clone(), outs[-1]["combined"].clone(), hot.kernel.amp_observation_buffer.clone())
        if fused:
            xs = hot.kernel.disc_input  # [N, padded / 32, 2, 32] plane blocks: column c = block c // 32, lane c % 32
            flat = xs.permute(0, 2, 1, 3).reshape(xs.shape[0], 2, -1)  # [N, plane, padded]
            assert flat.shape[2] == kd or float(flat[:, :, kd:].abs().max()) == 0.0  # padding untouched
            assert float(flat[:, 0, :kd].abs().max()) > 0.0
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)


def test_graph_replay_matches_eager():
    """The whole env-step captured as one hipGraph replays to the same bits as eager launches."""
    import contextlib
    import io

    from humanoid_amp_amd.workloads import WORKLOADS, HotPath

    res = {}
    for graph in (False, True):
        with contextlib.redirect_stdout(io.StringIO()):
            hot = HotPath(WORKLOADS["g1_walk"], 4096, "cuda:0", seed=8)
        if graph:
            hot.capture(warmup=2)   # 2 warm-up steps + the captured one ...
            n = 3
        else:
            n = 6                   # ... so eager runs 3 more to reach the same history state
        for _ in range(n):
            out = hot.step()
        hot.synchronize()
        res[graph] = (out["style"].clone(), out["combined"].clone(), hot.kernel.amp_observation_buffer.clone(),
                      hot.kernel.reset_ids.clone(), hot.kernel.policy_obs.clone())
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("precision,bar", [("f16x3", 1e-6), ("f32", 1e-6)])
@pytest.mark.parametrize("in_dim,rows", [(166, 4096), (830, 777), (162, 70000), (166, 20000), (166, 30001), (830, 49153),
                                         (166, 2500), (166, 3500), (830, 5000), (130, 30001), (142, 65536), (128, 33000)])
def test_gemm_engines_vs_fp64(precision, bar, in_dim, rows):
    """Both GEMM engines (fp16-split default, fp32 MFMA) must be as accurate as an fp32 forward: <= 1e-6 against the
    fp64 evaluation of the same fp32 weights on O(1) logits, a tenth of the path's 1e-5 budget.  The row counts walk
    through every kernel selection of the fp16 engine: 64 x 64 register-staged tiles (777, 2 500), the 4 096-env plan --
    LDS-DMA 128 x 128 layer 1 + 64 x 128 layer 2 on the four-stage ring (3 500 with a ragged tile, 4 096) or on two stages
    (5 000: more tiles than CUs) --, 128 x 128 LDS-DMA (20 000),
    one 256 x 256 LDS-DMA launch with a ragged last tile (30 001), 32 768-row chunks with a short last chunk (49 153,
    70 000).  Checked rows: the first 2 048 and the LAST 2 048 (ragged tiles / last chunk)."""
    from humanoid_amp_amd.engine import AmpDiscriminator

    g = torch.Generator().manual_seed(in_dim + rows)
    w = odisc.make_weights(in_dim, seed=3)
    x = torch.randn(rows, in_dim, generator=g) * 1.5
    mean = torch.randn(in_dim, generator=g, dtype=torch.float64) * 0.2
    var = torch.rand(in_dim, generator=g, dtype=torch.float64) + 0.1
    task = torch.randn(rows, 1, generator=g)
    d = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0", running_mean=mean, running_variance=var,
                         task_reward_weight=0.5, style_reward_weight=0.5, precision=precision)
    out = d.style_reward(x.cuda(), task.cuda(), want_logits=True)
    sub = torch.cat([torch.arange(0, min(rows, 2048)), torch.arange(max(rows - 2048, 0), rows)]).unique()
    ref = odisc.forward(w, x[sub], mean, var, task=task[sub], task_w=0.5, style_w=0.5)
    with torch.no_grad():
        lg64 = odisc.logits(w, ref["scaled"], dtype=torch.float64)
    scale = max(1.0, float(lg64.abs().max()))
    assert float((out["logits"][sub].cpu().double() - lg64).abs().max()) <= bar * scale
    assert float((out["style"][sub].cpu() - ref["style"]).abs().max()) <= 2.5 * bar * scale + 1e-6
    assert float((out["combined"][sub].cpu() - ref["combined"]).abs().max()) <= 2.5 * bar * scale + 1e-6
    # prescaled entry point (host-built input in the layout the engine reports) == generic entry point
    from humanoid_amp_amd import _native as nat

    lay = d.input_layout()
    xs = torch.zeros(rows, lay.padded_dim)
    xs[:, :in_dim] = odisc.scale_states(x, mean, var)
    if lay.format == nat.AMP_DISC_INPUT_F16_BLOCKS:
        assert precision == "f16x3" and lay.plane_scale == 4096.0  # 4096 * clip(5) < 2^15
        v = xs * lay.plane_scale
        p0 = v.half()
        p1 = (v - p0.float()).half()
        # block layout: [rows, padded / 32, plane, 32]
        xs = torch.stack([p0.view(rows, -1, 32), p1.view(rows, -1, 32)], dim=2).contiguous()
    pre = d.style_reward_prescaled(xs.cuda(), task.cuda(), want_logits=True)
    assert float((pre["logits"] - out["logits"]).abs().max()) <= bar * scale  # host-scaled vs device-scaled input


def test_f16_engine_without_clamp_uses_dynamic_bound():
    """No scaler => nothing bounds the input: the fp16 engine takes the abs-max of the batch as the plane bound.
    Inputs spanning 1e-6 .. 1e4 must neither overflow fp16 nor lose the small rows."""
    from humanoid_amp_amd.engine import AmpDiscriminator

    g = torch.Generator().manual_seed(77)
    w = odisc.make_weights(166, seed=4)
    x = torch.randn(600, 166, generator=g)
    x[:200] *= 1e-6
    x[200:400] *= 30.0
    x[599] *= 1e4
    d = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0")
    out = d.style_reward(x.cuda(), want_logits=True)["logits"].cpu().double()
    with torch.no_grad():
        lg64 = odisc.logits(w, x, dtype=torch.float64)
        lg32 = odisc.logits(w, x, dtype=torch.float32).double()
    assert torch.isfinite(out).all()
    err, err32 = (out - lg64).abs(), (lg32 - lg64).abs()
    # per-tensor plane scales: the error floor is relative to the LARGEST row of the batch (documented in DESIGN.md);
    # rows of ordinary magnitude must match an fp32 forward
    scale = lg64.abs().clamp(min=1.0)
    assert float((err[:599] / scale[:599]).max()) <= 1e-6 + float((err32[:599] / scale[:599]).max())
    assert float(err[599] / scale[599]) <= 2e-6


def test_lds_dma_kernels_agree_with_the_fp32_engine_over_many_launches():
    """Short race screen of the LDS-DMA GEMM kernels (tools/race_screen.py is the long form: 3.8e5 launches clean): many
    launches at random ragged shard sizes, repeated on the same input, against the fp32-MFMA engine (register staging,
    a different kernel family).  A mis-ordered LDS-DMA read is a wrong TILE -- an error of order one."""
    from humanoid_amp_amd.engine import AmpDiscriminator

    w = odisc.make_weights(166, seed=8)
    kw = dict(running_mean=torch.zeros(166, dtype=torch.float64), running_variance=torch.ones(166, dtype=torch.float64))
    fast = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0", precision="f16x3", **kw)
    slow = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0", precision="f32", **kw)
    g = torch.Generator(device="cuda").manual_seed(1)
    sizes = torch.randint(24576, 80000, (40,), generator=torch.Generator().manual_seed(2)).tolist()
    for rows in sizes:
        x = torch.randn(rows, 166, device="cuda", generator=g) * 1.5
        ref = slow.style_reward(x, want_logits=True)["logits"]
        first = None
        for _ in range(4):
            got = fast.style_reward(x, want_logits=True)["logits"]
            assert float((got - ref).abs().max()) <= 5e-6
            first = got if first is None else first
            assert torch.equal(got, first)  # launches on the same input are bit-identical


@pytest.mark.parametrize("w1_scale,w2_scale,b_scale", [(50.0, 0.02, 10.0), (1e-3, 300.0, 1e-3), (700.0, 1.0, 100.0)])
@pytest.mark.parametrize("rows", [3000, 40000])
def test_f16_engine_plane_scales_follow_the_weights(w1_scale, w2_scale, b_scale, rows):
    """The fp16 planes are scaled from bounds on the weights / hidden layer: badly scaled layers (hidden activations of
    order 1e4 or 1e-3) must neither overflow fp16 nor lose accuracy -- relative 1e-6 against fp64, like an fp32 forward."""
    from humanoid_amp_amd.engine import AmpDiscriminator

    g = torch.Generator().manual_seed(int(w1_scale * 7 + rows))
    w = odisc.make_weights(166, seed=21)
    w = [(w[0][0] * w1_scale, w[0][1] * b_scale), (w[1][0] * w2_scale, w[1][1] * b_scale), w[2]]
    x = torch.randn(rows, 166, generator=g) * 2.0
    mean = torch.zeros(166, dtype=torch.float64)
    var = torch.ones(166, dtype=torch.float64)
    d = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0", running_mean=mean, running_variance=var)
    got = d.style_reward(x.cuda(), want_logits=True)["logits"].cpu().double()
    sub = torch.cat([torch.arange(0, 1024), torch.arange(rows - 1024, rows)])
    with torch.no_grad():
        xs = odisc.scale_states(x[sub], mean, var)
        lg64 = odisc.logits(w, xs, dtype=torch.float64)
        lg32 = odisc.logits(w, xs, dtype=torch.float32).double()
    assert torch.isfinite(got).all()
    scale = float(lg64.abs().max())
    err, err32 = float((got[sub] - lg64).abs().max()), float((lg32 - lg64).abs().max())
    assert err <= 2e-6 * scale + 1e-6, (err, err32, scale)


def test_set_weights_in_place():
    """amp_disc_set_weights: 12 load_state_dict-style updates of ONE handle with a scaler set, an EnvStepKernel layout
    attached and a trainer attached.  After every update the handle must equal a FRESH handle built from the same
    weights + scaler bit for bit (logits, style), the layout's device pointers must not move (no re-allocation: the
    fused-scaler launch keeps reading valid memory), and the attached trainer must see the new weights."""
    from humanoid_amp_amd.engine import AmpDiscriminator, AmpDiscriminatorTrainer, EnvStepConfig, EnvStepKernel

    in_dim, rows = 166, 700
    g = torch.Generator().manual_seed(77)
    mean = torch.randn(in_dim, generator=g, dtype=torch.float64) * 0.2
    var = torch.rand(in_dim, generator=g, dtype=torch.float64) + 0.1
    x = (torch.randn(rows, in_dim, generator=g) * 2.0).cuda()
    task = torch.randn(rows, 1, generator=g).cuda()
    dev = lambda w: [(p.cuda(), q.cuda()) for p, q in w]  # noqa: E731
    disc = AmpDiscriminator(dev(odisc.make_weights(in_dim, seed=0)), "cuda:0", running_mean=mean, running_variance=var,
                            task_reward_weight=0.5, style_reward_weight=0.5)
    kern = EnvStepKernel(EnvStepConfig(n_dof=29, num_amp_observations=2, max_episode_length=300), 64, "cuda:0")
    kern.attach_discriminator(disc)
    lay0 = disc.input_layout()
    ptrs0 = (lay0.mean_dev, lay0.den_dev, lay0.format, lay0.padded_dim)
    trainer = AmpDiscriminatorTrainer(disc, batch_size=256, running_mean=mean, running_variance=var, apply_update=False,
                                      update_scaler=False)
    handle0 = disc._handle.value
    for it in range(12):
        w = odisc.make_weights(in_dim, seed=100 + it)
        disc.set_weights(dev(w))
        assert disc._handle.value == handle0
        lay = disc.input_layout()
        assert (lay.mean_dev, lay.den_dev, lay.format, lay.padded_dim) == ptrs0, "scaler pointers / layout moved"
        got = disc.style_reward(x, task, want_logits=True)
        fresh = AmpDiscriminator(dev(w), "cuda:0", running_mean=mean, running_variance=var, task_reward_weight=0.5,
                                 style_reward_weight=0.5).style_reward(x, task, want_logits=True)
        for k in ("logits", "style", "combined"):
            assert torch.equal(got[k], fresh[k]), (it, k)
        # the scaler survived (a dropped scaler would feed raw inputs): compare with the oracle as well
        ref = odisc.forward(w, x.cpu(), mean, var, task=task.cpu(), task_w=0.5, style_w=0.5)
        assert float((got["style"].cpu() - ref["style"]).abs().max()) <= TOL
        for (tw, tb), (ow, ob) in zip(trainer.weights(), w):
            assert torch.equal(tw.cpu(), ow) and torch.equal(tb.cpu(), ob), "trainer does not see the new weights"
    # the attached trainer still steps on live memory
    b = x[:256]
    out = trainer.step(b, b.flip(0), b * 0.5)
    assert torch.isfinite(out["loss"]).item()
    with pytest.raises(ValueError):
        disc.set_weights(dev(odisc.make_weights(162, seed=1)))


@pytest.mark.parametrize("in_dim", [166, 830, 130, 142, 100])
def test_a_rows_logit_does_not_depend_on_the_kernel_plan(in_dim):
    """The same 1 500 rows scored inside batches that select every kernel plan of the fp16 engine -- register-staged 64 x 64
    tiles (2 000 rows), LDS-DMA 128 x 128 + 64 x 128 on the four-stage ring (3 500) and on two stages (5 000), 128 x 128 (9 000,
    20 000), 256 x 256 + 256 x 128 (14 000), and at in_dim 166 / 142 / 130 / 100 (6 / 5 / 5 / 4 k-blocks of activation fragments: the
    23-DoF D = 71 variant of configs[1] is K D = 142) the fused two-layer kernel: one ragged round (30 000), a full round + a
    column-split remainder (40 000: 32 768 + 7 232), two rounds (58 000) and two full rounds + a remainder (70 000); at in_dim 830
    256 x 256 tiles (30 000, ragged last tile moved up) and 32 768-row chunks -- give the
    same logits and style rewards bit for bit: every kernel of a layer issues the same MFMA shape in the same k order and reduces
    the output layer in the same canonical order."""
    from humanoid_amp_amd.engine import AmpDiscriminator

    g = torch.Generator().manual_seed(in_dim)
    w = odisc.make_weights(in_dim, seed=5)
    x = (torch.randn(70000, in_dim, generator=g) * 1.5).cuda()
    mean = torch.randn(in_dim, generator=g, dtype=torch.float64) * 0.2
    var = torch.rand(in_dim, generator=g, dtype=torch.float64) + 0.1
    d = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0", running_mean=mean, running_variance=var)
    ref = d.style_reward(x[:2000], want_logits=True)
    for rows in (3500, 5000, 9000, 14000, 20000, 30000, 40000, 58000, 70000):
        out = d.style_reward(x[:rows], want_logits=True)
        assert torch.equal(out["logits"][:1500], ref["logits"][:1500]), rows
        assert torch.equal(out["style"][:1500], ref["style"][:1500]), rows
    # ... and the rows at the END of a ragged batch (the last row tile is moved up to end at M) equal the same rows scored alone
    tail = d.style_reward(x[30000 - 1500:30000], want_logits=True)["logits"]
    assert torch.equal(d.style_reward(x[:30000], want_logits=True)["logits"][-1500:], tail)


@pytest.mark.parametrize("hidden,rows", [((384, 128), 900), ((640, 384), 7000), ((512, 256), 3500), ((512, 256), 40000),
                                         ((1024, 256), 9000), ((128, 128), 300), ((768, 512), 30000), ((256, 512), 25000)])
def test_other_hidden_widths_vs_fp64(hidden, rows):
    """Hidden widths other than the reference's 1024 / 512 (agents/*.yaml:31-39): multiples of 128 (the engine's granularity) that are not multiples of 256 stay on
    the register-staged kernels at every batch size (their 64 x 64 and 128 x 128 tiles, ragged last tiles), 256-multiples take the
    LDS-DMA plans with one or two column tiles in layer 2 -- all on the 16 x 16 x 32 MFMA layout; a 512-wide second layer over another
    first-layer width (768, 256) at >= 24 576 rows takes the fused two-layer kernel (24 / 8 k-blocks instead of 32).  Bar as in
    test_gemm_engines_vs_fp64: logits within 1e-6 * max(1, |logit|) of the fp64 evaluation."""
    from humanoid_amp_amd.engine import AmpDiscriminator

    in_dim = 166
    g = torch.Generator().manual_seed(rows + hidden[0])
    w = odisc.make_weights(in_dim, seed=9, hidden=hidden)
    x = torch.randn(rows, in_dim, generator=g) * 1.5
    mean = torch.randn(in_dim, generator=g, dtype=torch.float64) * 0.2
    var = torch.rand(in_dim, generator=g, dtype=torch.float64) + 0.1
    d = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0", running_mean=mean, running_variance=var)
    out = d.style_reward(x.cuda(), want_logits=True)
    sub = torch.cat([torch.arange(0, min(rows, 1024)), torch.arange(max(rows - 1024, 0), rows)]).unique()
    ref = odisc.forward(w, x[sub], mean, var)
    with torch.no_grad():
        lg64 = odisc.logits(w, ref["scaled"], dtype=torch.float64)
    scale = max(1.0, float(lg64.abs().max()))
    assert float((out["logits"][sub].cpu().double() - lg64).abs().max()) <= 1e-6 * scale
    assert float((out["style"][sub].cpu() - ref["style"]).abs().max()) <= 2.5e-6 * scale + 1e-6
    # ... and a row's logit does not depend on the batch it is scored in
    part = d.style_reward(x[:min(rows, 700)].cuda(), want_logits=True)["logits"]
    assert torch.equal(part, out["logits"][:part.shape[0]])


def test_set_plan_switches_the_fused_kernel_per_handle_without_changing_a_bit():
    """amp_disc_set_plan / amp_disc_plan_info: the per-handle override decides whether 30 000 rows take the one-launch two-layer
    kernel or the column-split two-kernel plan, plan_info reports what will run, and the logits are bit-identical either way."""
    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import AmpDiscriminator

    g = torch.Generator().manual_seed(11)
    w = odisc.make_weights(166, seed=5)
    x = (torch.randn(30000, 166, generator=g) * 1.5).cuda()
    d = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0", running_mean=torch.zeros(166, dtype=torch.float64),
                         running_variance=torch.ones(166, dtype=torch.float64))
    auto = d.plan_info(30000)
    assert auto["fused_rows"] == 30000 and auto["precision"] == "f16x3" and auto["cu_count"] > 0
    with nat.KernelTrace(capacity=64) as tr:
        fused = d.style_reward(x, want_logits=True)["logits"]
    assert "disc_mlp_fused_kernel" in tr.summary()
    d.set_plan(fused=False)
    off = d.plan_info(30000)
    assert off["fused_rows"] == 0 and off["chunk_rows"] == 30000 and off["plan_name"].startswith("LDS-DMA 256x256")
    with nat.KernelTrace(capacity=64) as tr:
        split = d.style_reward(x, want_logits=True)["logits"]
    assert "disc_mlp_fused_kernel" not in tr.summary() and torch.equal(split, fused)
    d.set_plan(fused=True, fused_min_rows=128)           # the threshold moved down: 1 000 rows take the fused kernel too
    assert d.plan_info(1000)["fused_rows"] == 1000
    assert torch.equal(d.style_reward(x[:1000], want_logits=True)["logits"], fused[:1000])
    d.set_plan()
    assert d.plan_info(1000)["fused_rows"] == 0 and d.plan_info(30000)["fused_rows"] == 30000
    assert d.plan_info(4096)["plan_name"] == "LDS-DMA 128x128 + 64x128" and d.plan_info(100)["plan"] == 0
    with pytest.raises(nat.AmpEngineError):
        d.set_plan(fused=True, fused_min_rows=5)


def test_raw_row_input_of_the_fused_kernel_is_bit_identical_to_the_plane_input():
    """Where the whole batch takes the one-launch two-layer kernel it reads the fp32 observation rows itself (scaler, clamp and
    plane split of the 48 elements a lane holds: `disc_mlp_fused_kernel<KX, true>`), so neither a scaler pass nor the env step's
    fused scaler writes a scaled copy.  Same arithmetic, same bits: (1) style_reward on 32 768 rows (raw rows) == the same rows
    through the plane input (the fused plan switched off and on with a threshold that makes the batch a partial round is not needed:
    the two-kernel plans consume planes); (2) the hot path at 32 768 envs through amp_hot_step's raw-row branch == the same steps
    with the env step's fused scaler feeding plane blocks (one_call=False keeps the attached layout), over K + 2 steps; (3) odd row
    strides / unaligned rows fall back to the scaler pass and still agree."""
    import contextlib
    import io

    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.engine import AmpDiscriminator
    from humanoid_amp_amd.workloads import WORKLOADS, HotPath

    g = torch.Generator().manual_seed(21)
    w = odisc.make_weights(166, seed=6)
    x = (torch.randn(32768, 166, generator=g) * 1.5).cuda()
    mean = torch.randn(166, generator=g, dtype=torch.float64) * 0.2
    var = torch.rand(166, generator=g, dtype=torch.float64) + 0.1
    d = AmpDiscriminator([(p.cuda(), q.cuda()) for p, q in w], "cuda:0", running_mean=mean, running_variance=var)
    assert d.plan_info(32768)["raw_input"] and not d.plan_info(8192)["raw_input"] and not d.plan_info(40000)["raw_input"]
    with nat.KernelTrace(capacity=64) as tr:
        raw = d.style_reward(x, want_logits=True)
    assert "disc_scale_split_kernel" not in tr.summary() and "disc_mlp_fused_kernel" in tr.summary()
    d.set_plan(fused=False)
    with nat.KernelTrace(capacity=64) as tr:
        planes = d.style_reward(x, want_logits=True)
    assert "disc_scale_split_kernel" in tr.summary()
    assert torch.equal(raw["logits"], planes["logits"]) and torch.equal(raw["style"], planes["style"])
    d.set_plan()
    wide = torch.zeros(32768, 167, device="cuda")       # odd row stride: 8-B alignment of the rows is lost -> scaler pass
    wide[:, :166] = x
    with nat.KernelTrace(capacity=64) as tr:
        odd = d.style_reward(wide[:, :166], want_logits=True)
    assert "disc_scale_split_kernel" in tr.summary() and torch.equal(odd["logits"], raw["logits"])

    res = {}
    for one_call in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            hot = HotPath(WORKLOADS["g1_walk"], 32768, "cuda:0", seed=4, state_sets=2, one_call=one_call)
        assert hot.raw_rows == one_call and (hot.kernel.disc_input is None) == one_call
        kd = hot.spec.K * hot.spec.D
        g = torch.Generator().manual_seed(22)           # the same scaler for both runs
        hot.disc.set_scaler(torch.randn(kd, generator=g, dtype=torch.float64) * 0.3, torch.rand(kd, generator=g, dtype=torch.float64) + 0.2)
        if not one_call:
            hot.kernel.attach_discriminator(hot.disc)
        rec = []
        for _ in range(hot.spec.K + 2):
            out = hot.step()
            hot.synchronize()
            n = int(hot.kernel.reset_count.item())
            rec.append((out["style"].clone(), out["combined"].clone(), hot.kernel.reset_ids[:n].clone()))
        res[one_call] = (rec, hot.kernel.amp_observation_buffer.clone(), hot.kernel.policy_obs.clone())
    for a, b in zip(res[True][0], res[False][0]):
        assert all(torch.equal(u, v) for u, v in zip(a, b))
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])


@pytest.mark.parametrize("envs", [32768, 5000])
def test_style_reward_with_the_compaction_on_its_tail_equals_the_two_calls(envs):
    """amp_disc_style_reward_compact (AmpDiscriminator.style_reward(compact=kernel)): the reset-id compaction rides on the style
    reward's tail launch -- on the raw-row fused plan (32 768 rows) and on the fallback (5 000 rows: compaction, then the ordinary
    call).  Reset ids, count, logits, style and combined rewards equal the two separate calls bit for bit."""
    import contextlib
    import io

    from humanoid_amp_amd import _native as nat
    from humanoid_amp_amd.workloads import WORKLOADS, HotPath

    with contextlib.redirect_stdout(io.StringIO()):
        hot = HotPath(WORKLOADS["g1_walk"], envs, "cuda:0", seed=6, fused_scaler=False)
    k, d = hot.kernel, hot.disc
    k.launch(nat.AMP_PHASE_ALL, key_body_indexes=[0, 1, 2, 3], **hot._sim)
    amp = k.amp_observation_buffer.view(envs, -1)
    ids, count = k.compact_resets()
    want_ids, want_n = ids.clone(), int(count.item())
    want = d.style_reward(amp, k.reward, want_logits=True)
    k.reset_ids.fill_(-1)
    k.reset_count.zero_()
    with nat.KernelTrace(capacity=64) as tr:
        got = d.style_reward(amp, k.reward, want_logits=True, compact=k)
    torch.cuda.synchronize()
    names = tr.summary()
    assert ("step_tail_kernel" in names) == (envs == 32768) and ("disc_scale_split_kernel" in names) == (envs != 32768)
    assert int(k.reset_count.item()) == want_n > 0 and torch.equal(k.reset_ids[:want_n], want_ids[:want_n])
    for key in ("logits", "style", "combined"):
        assert torch.equal(got[key], want[key]), key
