"""The fused PPO minibatch update (include/dyros_ppo.h, isaacgymdyros_amd/ppo_update.py) against torch's autograd running the same
update under autocast with GradScaler and two Adam optimisers -- the form of the reference's calc_gradients
(learning/rl_games_custom/a2c_continuous_seperate.py:108-193), which examples/ppo_consumer.py keeps as its other path."""
import copy
import importlib.util
import math
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ppo():
    spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _autograd_update(ppo, net, opt_a, opt_c, scaler, c, obs, act, nlp_old, mu_old, adv, ret):
    """One minibatch update as examples/ppo_consumer.py::minibatch_update does it; returns (a_loss, c_loss, clip fraction, kl,
    unscaled gradients before clipping by parameter name)."""
    with torch.autocast("cuda", dtype=torch.float16):
        mu, logstd, value = net(obs)
        sigma = torch.exp(logstd)
        nlp = ppo.neglogp(act, mu, sigma, logstd)
        a_loss, cf = ppo.actor_loss(nlp_old, nlp, adv, c["e_clip"])
        c_loss = ppo.critic_loss(None, value, c["e_clip"], ret, False)
        al, cl = a_loss.mean(), c_loss.mean()
        loss = al + 0.5 * cl * c["critic_coef"]
    for p in net.parameters():
        p.grad = None
    scaler.scale(loss).backward()
    scaler.unscale_(opt_a); scaler.unscale_(opt_c)
    grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    torch.nn.utils.clip_grad_norm_(net.actor_parameters(), c["grad_norm"])
    scaler.step(opt_a); scaler.step(opt_c); scaler.update()
    with torch.no_grad():
        kl = ppo.policy_kl(mu.detach().float(), sigma.detach().float(), mu_old, torch.exp(net.sigma).expand_as(mu))
    return float(al.detach()), float(cl.detach()), float(cf), float(kl), grads


def _lively(net):
    with torch.no_grad():          # (the yaml's gain 0.01 makes every activation tiny: a livelier network is the harder test)
        for p in net.parameters():
            if p.requires_grad and p.dim() == 2:
                torch.nn.init.orthogonal_(p, gain=1.0)
            elif p.requires_grad:
                p.uniform_(-0.1, 0.1)


def _batch(ppo, ref, U, n, dev, seed=11):
    g = torch.Generator(device=dev).manual_seed(seed)
    obs = torch.randn(n, U.IN, generator=g, device=dev)
    with torch.no_grad():
        mu0, logstd, _ = ref(obs)
        sigma = torch.exp(logstd)
        act = mu0 + sigma * torch.randn(n, U.ACT, generator=g, device=dev)
        mu_old = mu0 + 0.1 * sigma * torch.randn(n, U.ACT, generator=g, device=dev)          # (an older policy: ratios on both sides of the clip range)
        nlp_old = ppo.neglogp(act, mu_old, sigma, logstd)
    adv = torch.randn(n, generator=g, device=dev)
    ret = torch.randn(n, generator=g, device=dev)
    return obs, act, nlp_old, mu_old, adv, ret


def _close16(a, b, ulps=3.0, floor=0.0):
    """fp16 results of the same sums in another order: a few fp16 ulps of the entry, or of 1e-3 of the tensor's largest entry (or of
    `floor`: a layer's output is the GEMM's fp16 result plus the bias, rounded again -- where the two cancel, the first rounding is an
    error of the product's size, not of the sum's)"""
    a, b = a.float(), b.float()
    tol = ulps * 2.0 ** -10 * torch.clamp(torch.maximum(b.abs(), 1e-3 * b.abs().max()), min=floor)
    return bool(((a - b).abs() <= tol).all()), float(((a - b).abs() / tol).max())


@pytest.mark.gpu
@pytest.mark.parametrize("mfma", [True, False], ids=["mfma", "gemm"])
def test_fused_update_piece_by_piece_against_torch(mfma):
    """Every stage of the fused update against torch arithmetic on the SAME inputs (no chaos from samples that sit on the clip boundary):
    staging, the three layers, the loss kernel's output gradient against autograd on the heads' outputs, the five backward GEMMs and
    masks, gradient statistics, the Adam step."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    torch.manual_seed(3)
    dev = "cuda:0"
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    _lively(net)
    ref = copy.deepcopy(net)
    B, nmb = 4096, 2
    fused = U.FusedPpoUpdate(net, c, B, nmb, dev, mfma=mfma)
    lr_a, lr_c = 3e-5, 5e-5
    fused.set_learning_rates(lr_a, lr_c)
    obs, act, nlp_old, mu_old, adv, ret = _batch(ppo, ref, U, B * nmb, dev)
    fused.bind_batch(obs, act, nlp_old, mu_old, adv, ret)
    for i in range(nmb):
        sl = slice(i * B, (i + 1) * B)
        p16, p, m0, v0 = fused.p16.clone(), fused.p.clone(), fused.m.clone(), fused.v.clone()
        W = {k: t.clone().float() for k, t in fused.views16.items()}
        scale = float(fused.state[U.K["DWP_S_SCALE"]])
        fused.update()
        torch.cuda.synchronize()
        lg = fused.logged().cpu().tolist()
        assert lg[7] == 0.0 and lg[6] == scale
        # forward
        assert torch.equal(fused.x16[:, :U.IN], obs[sl].half()) and float(fused.x16[:, U.IN:].abs().max()) == 0.0
        x = fused.x16.float()
        h1 = torch.relu(torch.matmul(x, W["W1"].transpose(1, 2)) + W["b1"].unsqueeze(1))
        ok, worst = _close16(fused.h1, h1, floor=0.1); assert ok, ("h1", worst)
        h2 = torch.relu(torch.matmul(fused.h1.float(), W["W2"].transpose(1, 2)) + W["b2"].unsqueeze(1))
        ok, worst = _close16(fused.h2, h2, floor=0.1); assert ok, ("h2", worst)
        out = torch.matmul(fused.h2.float(), W["W3"].transpose(1, 2)) + W["b3"].unsqueeze(1)
        ok, worst = _close16(fused.out, out, floor=0.1); assert ok, ("out", worst)
        # the loss kernel against autograd on the heads' outputs as they are in the fused buffers
        mu = fused.out[0, :, :U.ACT].float().requires_grad_()
        val = fused.out[1, :, :1].float().requires_grad_()
        logstd = net.sigma.expand_as(mu)
        sigma = torch.exp(logstd)
        nlp = ppo.neglogp(act[sl], mu, sigma, logstd)
        a_loss, cf = ppo.actor_loss(nlp_old[sl], nlp, adv[sl], c["e_clip"])
        c_loss = ppo.critic_loss(None, val, c["e_clip"], ret[sl].unsqueeze(1), False)
        al, cl = a_loss.mean(), c_loss.mean()
        (scale * (al + 0.5 * cl * c["critic_coef"])).backward()
        assert lg[0] == pytest.approx(float(al.detach()), rel=1e-5, abs=1e-6) and lg[1] == pytest.approx(float(cl.detach()), rel=1e-5)
        assert lg[2] == pytest.approx(float(ppo.bound_loss(mu.detach()).mean()), rel=1e-5) and lg[3] == pytest.approx(float(cf), abs=0.5 / B) and 0.05 < float(cf) < 0.95
        assert lg[4] == pytest.approx(float(ppo.policy_kl(mu.detach(), sigma, mu_old[sl], sigma)), rel=1e-4)
        ok, worst = _close16(fused.dout[0, :, :U.ACT], mu.grad, ulps=1.5); assert ok, ("dmu", worst)
        ok, worst = _close16(fused.dout[1, :, :1], val.grad, ulps=1.5); assert ok, ("dvalue", worst)
        assert float(fused.dout[0, :, U.ACT:].abs().max()) == 0.0 and float(fused.dout[1, :, 1:].abs().max()) == 0.0
        # backward: from the fused output gradient
        d3 = fused.dout.float()
        ok, worst = _close16(fused.gviews["W3"], torch.matmul(d3.transpose(1, 2), fused.h2.float())); assert ok, ("gW3", worst)
        dz2 = (torch.matmul(d3, W["W3"]).half().float()) * (fused.h2 > 0)
        ok, worst = _close16(fused.dh2, dz2); assert ok, ("dz2", worst)
        ok, worst = _close16(fused.gviews["W2"], torch.matmul(fused.dh2.float().transpose(1, 2), fused.h1.float())); assert ok, ("gW2", worst)
        dz1 = (torch.matmul(fused.dh2.float(), W["W2"]).half().float()) * (fused.h1 > 0)
        ok, worst = _close16(fused.dh1, dz1); assert ok, ("dz1", worst)
        ok, worst = _close16(fused.gviews["W1"], torch.matmul(fused.dh1.float().transpose(1, 2), x)); assert ok, ("gW1", worst)
        # the optimiser: unscale, actor clip, Adam -- from the gradients in the fused buffers and bias gradients = column sums
        gb = torch.cat([fused.dh1.float().sum(1).reshape(-1), fused.dh2.float().sum(1).reshape(-1), d3.sum(1).reshape(-1)])
        gv = fused.gviews          # (fp16 from the library GEMMs, fp32 from dwp_wgrad)
        gfull = torch.cat([gv[n_].reshape(-1).float() for n_ in ("W1", "W2", "W3")] + [gb]) / scale
        actor = torch.zeros(U.NP, dtype=torch.bool, device=dev)
        o = 0
        for nelem in (U.NW1, U.NW2, U.NW3, U.NB1, U.NB2, U.NB3):
            actor[o:o + nelem // 2] = True
            o += nelem
        norm = float(gfull[actor].norm())
        assert lg[5] == pytest.approx(norm, rel=1e-4)
        gfull = torch.where(actor, gfull * min(1.0, c["grad_norm"] / (norm + 1e-6)), gfull)
        step = float(i + 1)
        m1 = m0 + (gfull - m0) * 0.1
        v1 = 0.999 * v0 + 0.001 * gfull * gfull
        lr = torch.where(actor, torch.tensor(lr_a, device=dev), torch.tensor(lr_c, device=dev))
        pn = p - (lr / (1 - 0.9 ** step)) * (m1 / (v1.sqrt() / math.sqrt(1 - 0.999 ** step) + 1e-8))
        assert float((fused.m - m1).abs().max()) <= 1e-5 * float(m1.abs().max()) + 1e-12
        assert float((fused.v - v1).abs().max()) <= 1e-4 * float(v1.abs().max()) + 1e-20
        assert float((fused.p - pn).abs().max()) <= 0.02 * lr_a          # (bias sums are fp32 atomics in another order: a moment within rounding of zero)
        assert torch.equal(fused.p16, fused.p.half())          # (the fp16 copy the next forward reads: the stored master, rounded)
    assert float(fused.views["W3"][0, U.ACT:].abs().max()) == 0.0 and float(fused.views["b3"][1, 1:].abs().max()) == 0.0          # (padding rows stay zero)
    assert float(fused.views["W1"][:, :, U.IN:].abs().max()) == 0.0 and float(fused.gviews["W1"][:, :, U.IN:].abs().max()) == 0.0
    assert float(fused.state[U.K["DWP_S_MB"]]) == 0.0 and fused.state[U.K["DWP_S_STEP"]:U.K["DWP_S_STEP"] + 2].tolist() == [2.0, 2.0]


@pytest.mark.gpu
@pytest.mark.parametrize("mfma", [True, False], ids=["mfma", "gemm"])
def test_fused_update_tracks_the_autograd_update(mfma):
    """End to end against the autograd path under autocast (other GEMM tilings, so a handful of the 4096 samples land on the other side
    of the clip boundary: gradients agree in the norm, not entry by entry), and the loss scale moves alike."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    torch.manual_seed(3)
    dev = "cuda:0"
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    _lively(net)
    ref = copy.deepcopy(net)
    B, nmb = 4096, 3
    fused = U.FusedPpoUpdate(net, c, B, nmb, dev, mfma=mfma)
    lr_a, lr_c = 3e-5, 5e-5
    fused.set_learning_rates(lr_a, lr_c)
    opt_a = torch.optim.Adam(ref.actor_parameters(), lr=lr_a, eps=1e-8)
    opt_c = torch.optim.Adam(ref.critic_parameters(), lr=lr_c, eps=1e-8)
    scaler = torch.amp.GradScaler("cuda")
    obs, act, nlp_old, mu_old, adv, ret = _batch(ppo, ref, U, B * nmb, dev)
    fused.bind_batch(obs, act, nlp_old, mu_old, adv, ret)
    names = {"W1": ("actor_mlp.0.weight", "critic_mlp.0.weight"), "W2": ("actor_mlp.2.weight", "critic_mlp.2.weight"), "W3": ("mu.weight", "value.weight")}
    clean = 0
    for i in range(nmb):
        sl = slice(i * B, (i + 1) * B)
        with torch.no_grad():          # (every step starts from the same parameters)
            for pr, pf in zip(ref.parameters(), net.parameters()):
                pr.copy_(pf)
        al, cl, cf, kl, grads = _autograd_update(ppo, ref, opt_a, opt_c, scaler, c, obs[sl], act[sl], nlp_old[sl], mu_old[sl], adv[sl], ret[sl].unsqueeze(1))
        scale = float(fused.state[U.K["DWP_S_SCALE"]])
        fused.update()
        torch.cuda.synchronize()
        lg = fused.logged().cpu().tolist()
        # the two paths overflow fp16 together or not at all (the gradient at the heads' outputs is the same number in both), and then
        # skip, back off and count alike
        assert lg[6] == scale and float(fused.state[U.K["DWP_S_SCALE"]]) == scaler.get_scale()
        assert lg[7] == (1.0 if scaler.get_scale() < scale else 0.0)
        clean += lg[7] == 0.0
        if lg[7] != 0.0:
            continue
        assert lg[0] == pytest.approx(al, rel=5e-3, abs=5e-4) and lg[1] == pytest.approx(cl, rel=2e-3)
        assert lg[3] == pytest.approx(cf, abs=16.0 / B) and lg[4] == pytest.approx(kl, rel=5e-3)
        for w, (na, nc) in names.items():
            for k, nm in enumerate((na, nc)):
                gr = grads[nm]
                gf = fused.gviews[w][k, :gr.shape[0], :gr.shape[1]].float() / scale
                assert float((gf - gr).norm()) <= 0.25 * float(gr.norm()), (i, nm, float((gf - gr).norm()), float(gr.norm()))
        assert lg[5] == pytest.approx(math.sqrt(sum(float((grads[n_] ** 2).sum()) for n_ in grads if n_.startswith(("actor_mlp", "mu.")))), rel=3e-2)
    assert clean == nmb          # (this data does not overflow: every comparison above ran)


@pytest.mark.gpu
@pytest.mark.parametrize("mfma", [True, False], ids=["mfma", "gemm"])
def test_fused_update_skips_and_backs_off_on_overflow(mfma):
    """GradScaler's contract: a non-finite gradient in one net skips THAT optimiser's step, halves the scale, leaves the other net's
    step alone (separate unscale_ / step per optimiser, a2c_continuous_seperate.py:184-189)."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    torch.manual_seed(5)
    dev = "cuda:0"
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    B = 256
    fused = U.FusedPpoUpdate(net, c, B, 2, dev, mfma=mfma)
    fused.set_learning_rates(1e-3, 1e-3)
    g = torch.Generator(device=dev).manual_seed(1)
    obs = torch.randn(2 * B, U.IN, generator=g, device=dev)
    act = torch.randn(2 * B, U.ACT, generator=g, device=dev) * 0.1
    mu_old = torch.zeros(2 * B, U.ACT, device=dev)
    with torch.no_grad():
        nlp_old = ppo.neglogp(act, mu_old, torch.exp(net.sigma), net.sigma.expand_as(act))
    adv = torch.randn(2 * B, generator=g, device=dev)
    ret = torch.randn(2 * B, generator=g, device=dev)
    ret[:B] = 3.0e4          # value error x scale 65536 / B overflows fp16 in the critic's output gradient: inf in the critic only
    fused.bind_batch(obs, act, nlp_old, mu_old, adv, ret)
    p0 = fused.p.clone()
    fused.update()
    torch.cuda.synchronize()
    lg = fused.logged().cpu().tolist()
    assert lg[7] == 1.0 and float(fused.state[U.K["DWP_S_SCALE"]]) == 32768.0
    assert fused.state[U.K["DWP_S_STEP"]:U.K["DWP_S_STEP"] + 2].tolist() == [1.0, 0.0]
    moved = (fused.p != p0)
    crit = torch.zeros_like(moved)
    o = 0
    for nelem in (U.NW1, U.NW2, U.NW3, U.NB1, U.NB2, U.NB3):
        crit[o + nelem // 2:o + nelem] = True
        o += nelem
    assert not bool((moved & crit).any()) and bool((moved & ~crit).any())
    fused.update()          # minibatch 1 is clean: both step
    torch.cuda.synchronize()
    assert fused.logged()[7].item() == 0.0 and fused.state[U.K["DWP_S_STEP"]:U.K["DWP_S_STEP"] + 2].tolist() == [2.0, 1.0]


@pytest.mark.gpu
def test_consumer_trains_with_the_fused_update():
    """Two short epochs of the PPO consumer with the rollout and the fused update each replayed from a hipGraph: finite, and the losses
    are those of the autograd path on the same seed to within what fp16 GEMM tilings differ by (the first epoch's rollout is the same
    up to the last bit of the actions: same seed, same initial weights)."""
    ppo = _ppo()
    a = ppo.train(256, epochs=2, horizon=16, device="cuda:0", log=lambda *_: None, graph_rollout=True, graph_update=True)
    b = ppo.train(256, epochs=2, horizon=16, device="cuda:0", log=lambda *_: None, graph_rollout=True, fused_update=True)
    for sa, sb in zip(a, b):
        assert all(math.isfinite(sb[k]) for k in ("a_loss", "c_loss", "kl", "mean_reward"))
    # (the fused path's rollout draws the same noise but forms sigma with expf instead of torch.exp: actions differ in the last bit, the
    #  trajectories a little after 16 steps)
    assert b[0]["mean_reward"] == pytest.approx(a[0]["mean_reward"], rel=1e-3)
    assert b[0]["c_loss"] == pytest.approx(a[0]["c_loss"], rel=2e-2) and b[0]["a_loss"] == pytest.approx(a[0]["a_loss"], rel=5e-2, abs=2e-3)


@pytest.mark.gpu
def test_consumer_graph_chains_do_not_change_the_epoch():
    """Several updates / rollout steps per replayed graph (UPD_CHAIN, ROLL_CHAIN of examples/ppo_consumer.py: the minibatch index, the env's step
    counter and the rollout buffers' row index live on the device) against one per graph: the same epoch (the rollout to the bit; the update
    up to the order of the bias gradients' fp32 atomics)."""
    ppo = _ppo()
    keep = ppo.UPD_CHAIN, ppo.ROLL_CHAIN
    try:
        ppo.UPD_CHAIN, ppo.ROLL_CHAIN = 10 ** 6, 1
        a = ppo.train(256, epochs=2, horizon=16, device="cuda:0", log=lambda *_: None, graph_rollout=True, fused_update=True)
        ppo.UPD_CHAIN, ppo.ROLL_CHAIN = 2, 4          # (5 updates per epoch: 2 eager, one chain of 2, one single)
        b = ppo.train(256, epochs=2, horizon=16, device="cuda:0", log=lambda *_: None, graph_rollout=True, fused_update=True)
    finally:
        ppo.UPD_CHAIN, ppo.ROLL_CHAIN = keep
    assert b[0]["mean_reward"] == a[0]["mean_reward"] and b[0]["mean_episode_length"] == a[0]["mean_episode_length"]
    for sa, sb in zip(a, b):
        assert sb["mean_reward"] == pytest.approx(sa["mean_reward"], rel=1e-3)
        assert sb["c_loss"] == pytest.approx(sa["c_loss"], rel=1e-2) and sb["a_loss"] == pytest.approx(sa["a_loss"], rel=2e-2, abs=1e-3)
        assert sb["kl"] == pytest.approx(sa["kl"], rel=2e-2, abs=1e-5)


@pytest.mark.gpu
def test_fused_update_resumes_from_its_state_dict():
    """Parameters written from outside + refresh_copies(), and the optimiser state through state_dict() / load_state_dict(): a second updater
    built from them continues bit for bit (the matrix-core form is deterministic: no float atomics on anything the parameters depend on
    except the bias gradients' buckets, whose adds commute in fp32 only up to rounding -- so biases are compared to rounding, weights exactly)."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    dev = "cuda:0"
    torch.manual_seed(9)
    nets = [ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev) for _ in range(2)]
    _lively(nets[0])
    B, nmb = 1024, 4
    fa = U.FusedPpoUpdate(nets[0], c, B, nmb, dev)
    fa.set_learning_rates(3e-5, 5e-5)
    batch = _batch(ppo, copy.deepcopy(nets[0]), U, B * nmb, dev)
    fa.bind_batch(*batch)
    fa.update(); fa.update()
    torch.cuda.synchronize()
    fb = U.FusedPpoUpdate(nets[1], c, B, nmb, dev)
    with torch.no_grad():
        for pb, pa in zip(nets[1].parameters(), nets[0].parameters()):
            pb.copy_(pa)
    fb.load_state_dict(fa.state_dict())          # (calls refresh_copies)
    fb.set_learning_rates(3e-5, 5e-5)
    fb.bind_batch(*batch)
    fb.state[U.K["DWP_S_MB"]] = 2.0
    assert torch.equal(fb.p16, fa.p16) and torch.equal(fb.p16t, fa.p16t)
    fa.update(); fb.update()
    torch.cuda.synchronize()
    assert torch.equal(fa.out, fb.out) and torch.equal(fa.dout, fb.dout) and torch.equal(fa.g32, fb.g32)
    assert float((fa.p - fb.p).abs().max()) <= 1e-7 and fa.state[U.K["DWP_S_STEP"]:U.K["DWP_S_STEP"] + 2].tolist() == fb.state[U.K["DWP_S_STEP"]:U.K["DWP_S_STEP"] + 2].tolist()


@pytest.mark.gpu
def test_adam_finish_equals_adam_then_finish():
    """The merged last launch (dwp_adam_finish: Adam blocks that read loss scale / steps / flags from the partials buffer, an extra block that
    finishes the update meanwhile) against dwp_adam followed by dwp_finish, over clean updates and one that overflows the critic only: weights,
    moments, every copy of the weights and the whole state word for word; biases to the rounding of their buckets' atomic adds."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    dev = "cuda:0"
    torch.manual_seed(11)
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    _lively(net)
    B, nmb = 1024, 4
    fa = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, merged_tail=False)
    fb = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, split_tail=True)
    batch = list(_batch(ppo, copy.deepcopy(net), U, B * nmb, dev))
    batch[5] = batch[5].clone()
    batch[5][2 * B:3 * B] = 3.0e4          # (returns of the third minibatch: value error x scale / B overflows fp16 in the critic only)
    for f in (fa, fb):
        f.set_learning_rates(3e-5, 5e-5)
        f.bind_batch(*batch)
    skipped = []
    for _ in range(nmb + 1):
        fa.update(); fb.update()
        torch.cuda.synchronize()
        skipped.append(fa.logged()[7].item())
        nw = U.NWT
        assert torch.equal(fa.p[:nw], fb.p[:nw]) and torch.equal(fa.m[:nw], fb.m[:nw]) and torch.equal(fa.v[:nw], fb.v[:nw])
        assert torch.equal(fa.p16[:nw], fb.p16[:nw]) and torch.equal(fa.p16t, fb.p16t) and torch.equal(fa.p32f, fb.p32f)
        assert float((fa.p[nw:] - fb.p[nw:]).abs().max()) <= 1e-7 and float((fa.m[nw:] - fb.m[nw:]).abs().max()) <= 1e-6 * float(fa.m[nw:].abs().max()) + 1e-12
        sa, sb = fa.state.cpu(), fb.state.cpu()
        o = U.K["DWP_S_OUT"]
        keep = [i for i in range(U.K["DWP_S_WORDS"]) if not (o <= i < o + 6)]          # (the logged means and the norm: sums in another order)
        assert torch.equal(sa[keep], sb[keep]), (sa.tolist(), sb.tolist())
        assert torch.allclose(sa[o:o + 6], sb[o:o + 6], rtol=1e-5, atol=1e-7)
    assert skipped == [0.0, 0.0, 1.0, 0.0, 0.0]
    assert fa.state[U.K["DWP_S_STEP"]:U.K["DWP_S_STEP"] + 2].tolist() == [5.0, 4.0] and float(fa.state[U.K["DWP_S_SCALE"]]) == 32768.0


@pytest.mark.gpu
@pytest.mark.parametrize("sharded", [False, True], ids=["plain", "bucket_form"])
def test_stats_adam_finish_against_the_two_launches(sharded):
    """The three-launch update (merged_tail=True: dwp_stats_adam_finish: statistics, a grid barrier, clip + Adam + scaler in ONE launch) against dwp_grad_stats |
    dwp_adam_finish from the same parameters, over clean updates and one that overflows the critic only.  The critic's weights, moments and
    copies and the scaler's words are the same bits; the actor's are equal to the last place of the clip coefficient (the norm's partial sums
    are per Adam block in one form and per statistics block in the other); the barrier never timed out."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    dev = "cuda:0"
    torch.manual_seed(12)
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    _lively(net)
    B, nmb = 1024, 4
    kw = dict(collective=True) if sharded else {}
    fa = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, merged_tail=True, **kw)
    fb = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, merged_tail=False, **kw)
    assert fa.merged_tail and not fb.merged_tail
    batch = list(_batch(ppo, copy.deepcopy(net), U, B * nmb, dev))
    batch[5] = batch[5].clone()
    batch[5][2 * B:3 * B] = 3.0e4
    for f in (fa, fb):
        f.set_learning_rates(3e-5, 5e-5)
        f.bind_batch(*batch)
    skipped = []
    o = U.K["DWP_S_OUT"]
    for it in range(nmb + 1):
        # every update starts from the SAME parameters and moments (a last-place difference of the actor's would otherwise feed the next forward)
        with torch.no_grad():
            fb.p.copy_(fa.p); fb.m.copy_(fa.m); fb.v.copy_(fa.v); fb.p16.copy_(fa.p16); fb.p16t.copy_(fa.p16t); fb.p32f.copy_(fa.p32f)
        fa.update(); fb.update()
        torch.cuda.synchronize()
        skipped.append(fa.logged()[7].item())
        for name in ("W1", "W2", "W3"):
            assert torch.equal(fa.views[name][1], fb.views[name][1]) and torch.equal(fa.views16[name][1], fb.views16[name][1]), (it, name)
            da = (fa.views[name][0] - fb.views[name][0]).abs().max().item()
            assert da <= 2e-7, (it, name, da)          # (steps of ~3e-5 equal to ~1e-7 of themselves: at most the last place of a weight of size ~1)
        for name in ("b1", "b2", "b3"):          # (biases: to the rounding of their buckets' atomic adds, in both nets)
            assert float((fa.views[name] - fb.views[name]).abs().max()) <= 1e-7, (it, name)
        assert float((fa.m - fb.m).abs().max()) <= 1e-6 * float(fa.m.abs().max()) + 1e-12
        assert float((fa.v - fb.v).abs().max()) <= 1e-6 * float(fa.v.abs().max()) + 1e-12
        sa, sb = fa.state.cpu(), fb.state.cpu()
        keep = [i for i in range(U.K["DWP_S_WORDS"]) if not (o <= i < o + 6)]
        assert torch.equal(sa[keep], sb[keep]), (it, sa.tolist(), sb.tolist())
        assert torch.allclose(sa[o:o + 6], sb[o:o + 6], rtol=1e-5, atol=1e-7), (it, sa[o:o + 6].tolist(), sb[o:o + 6].tolist())
    assert skipped == [0.0, 0.0, 1.0, 0.0, 0.0]
    assert not fa.barrier_timed_out()
    assert int(fa.part.view(torch.int32)[641].item()) == nmb + 1


@pytest.mark.gpu
def test_stats_adam_finish_in_a_replayed_graph():
    """The merged launch captured in a hipGraph with the rest of the update and replayed: the barrier's arrival count carries over from one replay
    to the next (monotonic, nothing to reset), the minibatch index advances, the barrier never times out."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    dev = "cuda:0"
    torch.manual_seed(13)
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    _lively(net)
    B, nmb = 512, 4
    fa = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, merged_tail=True)
    fb = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, merged_tail=True)
    batch = list(_batch(ppo, copy.deepcopy(net), U, B * nmb, dev))
    for f in (fa, fb):
        f.bind_batch(*batch)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fa.update()          # (warm-up outside the capture)
    torch.cuda.current_stream().wait_stream(side)
    fb.update()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fa.update()
    for _ in range(11):
        g.replay(); fb.update()
    torch.cuda.synchronize()
    assert torch.equal(fa.state[U.K["DWP_S_STEP"]:U.K["DWP_S_STEP"] + 2], fb.state[U.K["DWP_S_STEP"]:U.K["DWP_S_STEP"] + 2])
    assert float((fa.p - fb.p).abs().max()) <= 1e-5          # (the bias gradients' atomic adds land in another order from run to run)
    assert fa.state[U.K["DWP_S_STEP"]].item() == 12.0 and not fa.barrier_timed_out() and int(fa.part.view(torch.int32)[641].item()) == 12


@pytest.mark.gpu
def test_policy_copy_once_per_epoch_equals_per_update():
    """policy_copy_per_update=False: the updates leave dwp_policy's fp32 operand-order copy alone, policy() refuses to read it while stale, and after
    sync_policy_copy() the copy -- and the policy's outputs -- are the same bits as with a copy rewritten by every update."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    dev = "cuda:0"
    torch.manual_seed(14)
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    _lively(net)
    B, nmb = 512, 4
    batch = list(_batch(ppo, copy.deepcopy(net), U, B * nmb, dev))
    obs = batch[0][:1000].contiguous()
    for merged in (False, True):
        fa = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, merged_tail=merged)
        fb = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, merged_tail=merged, policy_copy_per_update=False)
        for f in (fa, fb):
            f.bind_batch(*batch)
        before = fb.p32f.clone()
        for _ in range(3):
            with torch.no_grad():
                fb.p.copy_(fa.p); fb.m.copy_(fa.m); fb.v.copy_(fa.v); fb.p16.copy_(fa.p16); fb.p16t.copy_(fa.p16t)
            fa.update(); fb.update()
        torch.cuda.synchronize()
        assert torch.equal(fb.p32f, before) and not torch.equal(fa.p32f, before)
        with pytest.raises(RuntimeError):
            fb.policy(obs)
        with torch.no_grad():
            fb.p.copy_(fa.p)          # (bias gradients' atomic adds land in another order from run to run)
        fb.sync_policy_copy()
        torch.cuda.synchronize()
        assert torch.equal(fa.p32f, fb.p32f)
        (mu_a, v_a), (mu_b, v_b) = fa.policy(obs), fb.policy(obs)
        assert torch.equal(mu_a, mu_b) and torch.equal(v_a, v_b)


@pytest.mark.gpu
def test_gae_kernel_equals_the_reference_loop():
    """dwp_gae against `discount_values` (examples/ppo_consumer.py, restating a2c_common_dyros.py:485-500) on random rollouts with dones."""
    from isaacgymdyros_amd.ppo_update import gae
    ppo = _ppo()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(4)
    for H, N in ((128, 1000), (7, 3), (1, 64)):
        rew, val = torch.randn(H, N, 1, generator=g, device=dev), torch.randn(H, N, 1, generator=g, device=dev)
        done = (torch.rand(H, N, generator=g, device=dev) < 0.05).float()
        fd = (torch.rand(N, generator=g, device=dev) < 0.05).float()
        lv = torch.randn(N, 1, generator=g, device=dev)
        ref = ppo.discount_values(fd, lv, done, val, rew, 0.99, 0.95)
        got = gae(fd, lv, done, val, rew, 0.99, 0.95)
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-6, (H, N)


@pytest.mark.gpu
@pytest.mark.parametrize("env_major", [False, True, "half"], ids=["step_major", "env_major_obs", "env_major_fp16_obs"])
def test_rollout_recorder_equals_the_torch_bookkeeping(env_major):
    """dwp_rollout_pre / _post against the torch lines of examples/ppo_consumer.py::rollout_step (a2c_common_dyros.py:629-703) on the same draws;
    env_major: the observations recorded straight into the env-major flat batch (swap_and_flatten01 of :1080 as the rollout goes)."""
    from isaacgymdyros_amd.ppo_update import RolloutRecorder, ACT
    ppo = _ppo()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(2)
    H, N, NOBS = 4, 1000, 487
    mk = lambda *sh: torch.zeros(*sh, device=dev)
    mb = dict(obs=mk(H, N, NOBS), act=mk(H, N, ACT), neglogp=mk(H, N), val=mk(H, N, 1), rew=mk(H, N, 1), done=mk(H, N), mu=mk(H, N, ACT))
    ref = {k: v.clone() for k, v in mb.items()}
    n = torch.zeros(1, dtype=torch.long, device=dev)
    logstd = torch.full((ACT,), -2.3, device=dev) + 0.1 * torch.randn(ACT, generator=g, device=dev)
    flat_obs = (torch.zeros(N * H, 512, device=dev, dtype=torch.float16) if env_major == "half" else torch.zeros(N * H, NOBS, device=dev)) if env_major else None
    if env_major:
        mb["obs"] = None
    rec = RolloutRecorder(mb, n, logstd, 0.5, 0.99, True, obs_env_major=flat_obs, num_obs=NOBS)
    terms, terms_ref = torch.zeros(15, device=dev), torch.zeros(15, device=dev)
    g_dones, g_obs = torch.zeros(N, device=dev), torch.randn(N, NOBS, generator=g, device=dev)
    for step in range(H):
        mu, value, noise = torch.randn(N, ACT, generator=g, device=dev), torch.randn(N, 1, generator=g, device=dev), torch.randn(N, ACT, generator=g, device=dev)
        obs_in, dones_in = g_obs.clone(), g_dones.clone()
        act = rec.pre(mu, value, noise, g_obs, g_dones)
        sigma = torch.exp(logstd)
        a = mu + sigma * noise
        for k, v in (("obs", obs_in), ("act", a), ("mu", mu), ("neglogp", ppo.neglogp(a, mu, sigma, logstd.expand_as(mu))), ("val", value), ("done", dones_in)):
            ref[k][step] = v
        assert float((act - torch.clamp(a, -1.0, 1.0)).abs().max()) <= 2e-7          # (mu + sigma * noise as one fused multiply-add, sigma by expf)
        rew, tout = torch.randn(N, generator=g, device=dev), (torch.rand(N, generator=g, device=dev) < 0.1).long()
        stacked, d = torch.randn(N, 15, generator=g, device=dev), (torch.rand(N, generator=g, device=dev) < 0.2).long()
        new_obs = torch.randn(N, NOBS, generator=g, device=dev)
        rec.post(rew, value, tout, stacked, d, new_obs, terms, g_dones, g_obs)
        ref["rew"][step] = rew.unsqueeze(1) * 0.5 + 0.99 * value * tout.unsqueeze(1).float()
        terms_ref += stacked.mean(0)
        torch.cuda.synchronize()
        assert torch.equal(g_dones, d.float()) and torch.equal(g_obs, new_obs)
        n += 1
    if env_major == "half":          # (fp16 rows of 512: the cast autocast gives the first Linear's input, zero padding)
        assert torch.equal(flat_obs[:, :NOBS], ref["obs"].transpose(0, 1).reshape(N * H, NOBS).half()) and float(flat_obs[:, NOBS:].abs().max()) == 0.0
    elif env_major:
        assert torch.equal(flat_obs, ref["obs"].transpose(0, 1).reshape(N * H, NOBS))
    for k in ("mu", "val", "done") if env_major else ("obs", "mu", "val", "done"):
        assert torch.equal(mb[k], ref[k]), k
    assert float((mb["act"] - ref["act"]).abs().max()) <= 5e-7
    assert float((mb["neglogp"] - ref["neglogp"]).abs().max()) <= 2e-5 * float(ref["neglogp"].abs().max())
    assert float((mb["rew"] - ref["rew"]).abs().max()) <= 1e-6 * float(ref["rew"].abs().max()) + 1e-7
    assert float((terms - terms_ref).abs().max()) <= 1e-5


@pytest.mark.gpu
def test_fp16_observation_batch_gives_the_same_update():
    """bind_batch with the observations as fp16 rows of 512 (what RolloutRecorder writes for the trainer) against the fp32 batch: the staged
    input, every activation, the output gradient and the weight-gradient partials are the same bits."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    dev = "cuda:0"
    torch.manual_seed(5)
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    _lively(net)
    B, nmb = 2048, 2
    obs, act, nlp_old, mu_old, adv, ret = _batch(ppo, copy.deepcopy(net), U, B * nmb, dev)
    obs16 = torch.zeros(B * nmb, U.INP, device=dev, dtype=torch.float16)
    obs16[:, :U.IN] = obs.half()
    fa = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev)
    fb = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev)
    fa.bind_batch(obs, act, nlp_old, mu_old, adv, ret)
    fb.bind_batch(obs16, act, nlp_old, mu_old, adv, ret)
    for _ in range(nmb):
        fa.update(); fb.update()
        torch.cuda.synchronize()
        for name in ("x16", "h1", "h2", "out", "dout", "dh2", "dh1", "g32"):
            assert torch.equal(getattr(fa, name), getattr(fb, name)), name
    with pytest.raises(ValueError):
        U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, mfma=False).bind_batch(obs16, act, nlp_old, mu_old, adv, ret)          # (the library-GEMM form stages fp32)


@pytest.mark.gpu
def test_policy_forward_equals_the_module_in_fp32():
    """dwp_policy (v_mfma_f32_16x16x4_f32, fp32 weights in operand order) against the torch module's fp32 forward; also after an update (dwp_adam keeps
    the operand-order copy) and after refresh_copies()."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    dev = "cuda:0"
    torch.manual_seed(12)
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    _lively(net)
    B, nmb = 1024, 2
    f = U.FusedPpoUpdate(net, c, B, nmb, dev)
    f.set_learning_rates(1e-3, 1e-3)
    g = torch.Generator(device=dev).manual_seed(3)
    obs = torch.randn(4096, U.IN, generator=g, device=dev)

    def check(obs=obs):
        with torch.no_grad():
            mu_t, _, v_t = net(obs)
        mu, v = f.policy(obs)
        torch.cuda.synchronize()
        assert float((mu - mu_t).abs().max()) <= 2e-5 * float(mu_t.abs().max()) + 1e-6, float((mu - mu_t).abs().max())
        assert float((v - v_t).abs().max()) <= 2e-5 * float(v_t.abs().max()) + 1e-6
    check()
    check(obs[:1000]); check(obs[:3]); check(obs[:33])          # (env counts that are not a multiple of the workgroup's 32 rows)
    f.bind_batch(*_batch(ppo, copy.deepcopy(net), U, B * nmb, dev))
    f.update(); f.update()
    check()          # (the parameters moved by 2 x lr; the module sees the masters, dwp_policy the copy dwp_adam kept)
    with torch.no_grad():
        for p_ in net.parameters():
            if p_.requires_grad:
                p_.mul_(1.01)
    f.refresh_copies()
    check()


@pytest.mark.gpu
def test_sharded_form_of_the_update_on_one_rank_is_bit_identical():
    """FusedPpoUpdate(collective=True): dwp_mlp | dwp_wgrad | dwp_grad_bucket | (all-reduce: nothing to do on one rank) | dwp_grad_stats and
    dwp_adam_finish on the ONE-slab bucket -- against the plain four launches from the same parameters on the same batch, over clean updates
    and one that overflows the critic only: weights, moments, every copy of the weights and the state word for word (1 / world = 1 is exact and
    the slabs are summed in dwp_grad_stats' order); biases to the rounding of their buckets' atomic adds, as everywhere."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    dev = "cuda:0"
    torch.manual_seed(13)
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    _lively(net)
    B, nmb = 1024, 4
    fa = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev)
    fb = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, collective=True)
    assert fb.bucket is not None and fb.bucket.numel() == U.NWT + U.NBT and fa.bucket is None
    batch = list(_batch(ppo, copy.deepcopy(net), U, B * nmb, dev))
    batch[5] = batch[5].clone()
    batch[5][2 * B:3 * B] = 3.0e4          # (returns of the third minibatch: the critic's output gradient overflows fp16)
    for f in (fa, fb):
        f.set_learning_rates(3e-5, 5e-5)
        f.bind_batch(*batch)
    skipped = []
    for _ in range(nmb + 1):
        fa.update(); fb.update()
        torch.cuda.synchronize()
        skipped.append(fb.logged()[7].item())
        nw = U.NWT
        if skipped[-1] == 0.0:          # (an overflowing update has inf / nan entries, which compare unequal to themselves)
            assert torch.equal(fa.g32, fb.g32)
            assert torch.equal(fb.bucket[:nw], ((fb.g32[0] + fb.g32[1]) + fb.g32[2]) + fb.g32[3])
        assert torch.equal(fa.p[:nw], fb.p[:nw]) and torch.equal(fa.m[:nw], fb.m[:nw]) and torch.equal(fa.v[:nw], fb.v[:nw])
        assert torch.equal(fa.p16[:nw], fb.p16[:nw]) and torch.equal(fa.p16t, fb.p16t) and torch.equal(fa.p32f, fb.p32f)
        assert float((fa.p[nw:] - fb.p[nw:]).abs().max()) <= 1e-7
        sa, sb = fa.state.cpu(), fb.state.cpu()
        o = U.K["DWP_S_OUT"]
        keep = [i for i in range(U.K["DWP_S_WORDS"]) if not (o <= i < o + 6)]
        assert torch.equal(sa[keep], sb[keep]), (sa.tolist(), sb.tolist())
        assert torch.allclose(sa[o:o + 6], sb[o:o + 6], rtol=1e-5, atol=1e-7)
    assert skipped == [0.0, 0.0, 1.0, 0.0, 0.0]
    # the bucket path also through the consumer: the collective's place captured inside the chain of updates, then as two replayed graphs
    a = ppo.train(num_envs=256, epochs=2, horizon=16, device=dev, log=lambda s: None, graph_rollout=True, fused_update=True)
    b = ppo.train(num_envs=256, epochs=2, horizon=16, device=dev, log=lambda s: None, graph_rollout=True, fused_update=True, fused_collective=True)
    ppo.GRAPH_COLLECTIVE = False          # (the fall-back when a capture refuses the collective: head graph | all-reduce | tail graph)
    b2 = ppo.train(num_envs=256, epochs=2, horizon=16, device=dev, log=lambda s: None, graph_rollout=True, fused_update=True, fused_collective=True)
    assert a[0]["mean_reward"] == b2[0]["mean_reward"] and all(abs(x["a_loss"] - y["a_loss"]) <= 1e-4 for x, y in zip(a, b2))
    # (the first rollout is the same bits; after it the bias gradients' atomic adds round in launch order, as between two runs of one path)
    assert a[0]["mean_reward"] == b[0]["mean_reward"]
    for x, y in zip(a, b):
        assert abs(x["mean_reward"] - y["mean_reward"]) <= 1e-4 and abs(x["a_loss"] - y["a_loss"]) <= 1e-4 and abs(x["kl"] - y["kl"]) <= 1e-4, (x, y)


@pytest.mark.gpu
def test_update_without_learning_rates_or_batch_is_refused():
    """a zero learning rate used to advance Adam's moments and step counts while no parameter moved, silently.  The rates now start at
    the cfg's; a cfg without them must be completed by set_learning_rates() before update()."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    dev = "cuda:0"
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    f = U.FusedPpoUpdate(copy.deepcopy(net), c, 64, 2, dev)
    assert f.state[U.K["DWP_S_LR"]:U.K["DWP_S_LR"] + 2].tolist() == pytest.approx([c["learning_rate"], c["critic_lr"]])
    with pytest.raises(RuntimeError, match="bind_batch"):
        f.update()
    c2 = {k: v for k, v in c.items() if k not in ("learning_rate", "critic_lr")}
    g = U.FusedPpoUpdate(copy.deepcopy(net), c2, 64, 2, dev)
    g.bind_batch(*_batch(ppo, copy.deepcopy(net), U, 128, dev))
    with pytest.raises(RuntimeError, match="learning rates"):
        g.update()
    g.set_learning_rates(1e-4, 1e-4)
    g.update()
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_rollout_recorder_refuses_bad_arguments_and_drops_rows_past_the_buffers():
    """every per-call tensor of pre / post is checked for dtype, size, contiguity and device before its address reaches a
    kernel -- a bool time-out mask (TocabiAMPLower's timeout_buf) would be read as int64, 8 x out of bounds --; and the device row counter is
    bounded by H inside the kernels: a replayed rollout graph whose caller forgot to rewind n records nothing instead of writing past every
    rollout buffer.  Also: 35 logged reward columns (a terrain curriculum's 15 + 20 types) are reduced, where 16 was the limit."""
    from isaacgymdyros_amd.ppo_update import RolloutRecorder, ACT
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(3)
    H, N, NOBS, COLS = 3, 512, 487, 35
    mk = lambda *sh: torch.zeros(*sh, device=dev)
    mb = dict(obs=mk(H, N, NOBS), act=mk(H, N, ACT), neglogp=mk(H, N), val=mk(H, N, 1), rew=mk(H, N, 1), done=mk(H, N), mu=mk(H, N, ACT))
    n = torch.zeros(1, dtype=torch.long, device=dev)
    logstd = torch.full((ACT,), -2.0, device=dev)
    rec = RolloutRecorder(mb, n, logstd, 1.0, 0.99, True)
    rnd = lambda *sh: torch.randn(*sh, generator=g, device=dev)
    mu, value, noise, obs, dones = rnd(N, ACT), rnd(N, 1), rnd(N, ACT), rnd(N, NOBS), mk(N)
    rew, stacked, new_obs = rnd(N), rnd(N, COLS), rnd(N, NOBS)
    tout, d = (torch.rand(N, generator=g, device=dev) < 0.1).long(), (torch.rand(N, generator=g, device=dev) < 0.2).long()
    terms, g_dones, g_obs = mk(COLS), mk(N), mk(N, NOBS)
    for bad in (dict(mu=mu[:, :12].contiguous()), dict(noise=noise.double()), dict(obs=obs[:, :480].contiguous()), dict(dones=dones.long()), dict(value=value.cpu())):
        with pytest.raises(ValueError):
            rec.pre(**dict(dict(mu=mu, value=value, noise=noise, obs=obs, dones=dones), **bad))
    ok = dict(rew=rew, value=value, time_outs=tout, stacked=stacked, done_buf=d, new_obs=new_obs, terms=terms, g_dones=g_dones, g_obs=g_obs)
    for bad in (dict(time_outs=tout.bool()), dict(done_buf=d.int()), dict(stacked=stacked.t()), dict(terms=mk(COLS + 1)), dict(terms=mk(65)), dict(rew=rew[:-1]), dict(g_obs=g_obs.half())):
        with pytest.raises(ValueError):
            rec.post(**dict(ok, **bad))
    with pytest.raises(ValueError):
        RolloutRecorder(dict(mb, act=mk(H, N, 12)), n, logstd, 1.0, 0.99, True)
    with pytest.raises(ValueError):
        RolloutRecorder(mb, n.int(), logstd, 1.0, 0.99, True)
    # H good steps, then two more without a rewind
    for step in range(H + 2):
        act = rec.pre(mu, value, noise, obs, dones)
        rec.post(**ok)
        n += 1
        if step == H - 1:
            torch.cuda.synchronize()
            assert rec.rows() == H
            snap = {k: v.clone() for k, v in mb.items()}
            t_snap = terms.clone()
    torch.cuda.synchronize()
    assert float((terms[:COLS] - H * stacked.mean(0)).abs().max()) <= 1e-5 and torch.equal(terms, t_snap)          # (35 columns; nothing added past H)
    for k in mb:
        assert torch.equal(mb[k], snap[k]), k
    assert torch.equal(mb["obs"][H - 1], obs) and float(mb["rew"].abs().max()) > 0
    assert float((act - torch.clamp(mu + torch.exp(logstd) * noise, -1.0, 1.0)).abs().max()) <= 2e-7          # (the env still gets its action)
    assert torch.equal(g_obs, new_obs) and torch.equal(g_dones, d.float())
    with pytest.raises(RuntimeError, match="row counter"):
        rec.rows()
    n.fill_(-1)
    rec.pre(mu, value, noise, obs, dones); rec.post(**ok)
    torch.cuda.synchronize()
    for k in mb:
        assert torch.equal(mb[k], snap[k]), k


@pytest.mark.gpu
def test_fp16_grads_switch_gives_the_weight_gradients_autocasts_overflow():
    """the four-launch form keeps the weight gradients as fp32 sums, so with gradients beyond the fp16 range its found_inf stays
    down where a backward under autocast (and the library-GEMM form, whose GEMMs write fp16) overflows at 65 504.  FusedPpoUpdate(fp16_grads=True)
    rounds the summed weight gradients through fp16 before the statistics and Adam (DWP_S_G16): same skip, same back-off as the GEMM form.  A loss
    scale of 2^21 over a minibatch of 4096 keeps the output gradients within fp16 per sample and puts their sums over the samples beyond 65 504."""
    from isaacgymdyros_amd import ppo_update as U
    ppo = _ppo()
    c = dict(ppo.TRAIN_CFG["config"])
    dev = "cuda:0"
    torch.manual_seed(17)
    net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
    _lively(net)
    B, nmb = 4096, 2
    S0 = float(2 ** 21)
    batch = _batch(ppo, copy.deepcopy(net), U, B * nmb, dev)
    fs = {"plain": U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev), "fp16": U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, fp16_grads=True),
          "gemm": U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, mfma=False)}
    for f in fs.values():
        f.state[U.K["DWP_S_SCALE"]] = S0
        f.bind_batch(*batch)
        f.update()
    torch.cuda.synchronize()
    lg = {k: f.logged().cpu().tolist() for k, f in fs.items()}
    steps = {k: f.state[U.K["DWP_S_STEP"]:U.K["DWP_S_STEP"] + 2].tolist() for k, f in fs.items()}
    scale = {k: float(f.state[U.K["DWP_S_SCALE"]]) for k, f in fs.items()}
    # the CRITIC is the clean case: its output gradient scale * (value - return) / B is in range for every sample (the actor's, (a - mu) / sigma^2
    # times the advantage, has tails that overflow by themselves at this scale -- in every form alike)
    assert torch.isfinite(fs["plain"].dout[1].float()).all()
    assert steps["gemm"][1] == 0.0 and lg["gemm"][7] == 1.0 and scale["gemm"] == S0 / 2, (lg["gemm"], steps, scale)          # autocast's arithmetic overflows in the weight gradients ...
    assert steps["plain"][1] == 1.0, steps          # ... the fp32 sums do not: the critic steps ...
    assert steps["fp16"] == steps["gemm"] and lg["fp16"][7] == 1.0 and scale["fp16"] == scale["gemm"]          # ... and with the switch they do, net by net
    # back in range (the halved scale is still too high here, so lower it): the switch then only rounds -- both mfma forms take the step
    for f in fs.values():
        f.state[U.K["DWP_S_SCALE"]] = 1024.0
        f.update()
    torch.cuda.synchronize()
    assert all(f.logged()[7].item() == 0.0 for f in fs.values())
    assert all(bool(torch.isfinite(f.p).all()) for f in fs.values())
    # (the switch by itself, from equal parameters: the rounded gradients differ from the fp32 sums by 2^-11 relative -- a different, close step)
    fa, fb = U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev), U.FusedPpoUpdate(copy.deepcopy(net), c, B, nmb, dev, fp16_grads=True)
    for f in (fa, fb):
        f.state[U.K["DWP_S_SCALE"]] = 256.0
        f.bind_batch(*batch)
        f.update()
    torch.cuda.synchronize()
    d = (fa.p - fb.p).abs().max().item()
    assert fa.logged()[7].item() == 0.0 and fb.logged()[7].item() == 0.0 and 0.0 < d < 1e-4, d
