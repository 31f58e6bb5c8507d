"""Fused PPO minibatch-gradient kernels (csrc/ppo_train.hip) against PyTorch fp32 autograd of the reference's loss
(elegantrl/agent.py:637-655) on the same minibatch.  A floating-point kernel: the checker is a plain torch fp32
implementation on the same device, with tolerances stated per assertion."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(kind, md, D, seed):
    from pime_amd.elegantrl.net import ActorPPO, CriticAdv
    from pime_amd.elegantrl.net_residual import ActorResidualIntegratorModularPPO, ActorResidualPPO
    torch.manual_seed(seed)
    cri = CriticAdv(D, md).to(DEV)
    if kind == "modular":
        act = ActorResidualIntegratorModularPPO(md, D, 1, 1).to(DEV)
    elif kind == "resid":
        act = ActorResidualPPO(md, D, 1).to(DEV)
    else:
        act = ActorPPO(md, D, 1).to(DEV)
    with torch.no_grad():
        act.net[-1].weight.mul_(6.0)
        act.a_std_log.fill_(-0.3)
    return act, cri


def _data(L, D, act, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    scale = torch.tensor(([3., 3., 8., 1.] * 8)[:D])
    shift0 = torch.tensor(([7., 7., 0., 0.] * 8)[:D])
    state = (torch.randn(L, D, generator=g) * scale + shift0).to(DEV)
    with torch.no_grad():
        mean = act.mean(state)
        std = act.a_std_log.exp()
        noise = torch.randn(L, 1, generator=g).to(DEV)
        action = mean + noise * std
        logprob = act.old_logprob(noise)
        # move a third of the old log-probs so that both clip branches and both advantage signs occur
        shift = (torch.rand(L, generator=g).to(DEV) - 0.5) * 0.8
        logprob = logprob + torch.where(torch.rand(L, generator=g).to(DEV) < 0.35, shift, torch.zeros_like(shift))
    adv = torch.randn(L, generator=g).to(DEV)
    r_sum = (torch.randn(L, generator=g) * 30 - 40).to(DEV)
    return state, action, logprob, adv, r_sum


def _torch_grads(act, cri, state, action, logprob, adv, r_sum, idx, clip, lam):
    for p in list(act.parameters()) + list(cri.parameters()):
        p.grad = None
    s, a, lp, ad, rs = state[idx], action[idx], logprob[idx], adv[idx], r_sum[idx]
    new_lp = act.compute_logprob(s, a)
    ratio = (new_lp - lp).exp()
    sur = torch.min(ad * ratio, ad * ratio.clamp(1 - clip, 1 + clip))
    ent = (new_lp.exp() * new_lp).mean()
    obj_a = -sur.mean() + ent * lam
    obj_c = torch.nn.functional.smooth_l1_loss(cri(s).squeeze(1), rs)
    scale = 1.0 / (rs.std() + 1e-5)
    (obj_a + obj_c * scale).backward()
    grads = {n: p.grad.clone() for n, p in list(act.named_parameters()) + [("cri." + k, v) for k, v in cri.named_parameters()]
             if p.grad is not None}
    return grads, float(-sur.sum().detach()), float((new_lp.exp() * new_lp).sum().detach()), float(obj_c.detach()) * len(idx), scale.reshape(1)


@pytest.mark.parametrize("kind,md,D,B", [("modular", 128, 3, 4096), ("modular", 128, 3, 1000), ("modular", 64, 4, 2048),
                                         ("resid", 128, 3, 4096), ("resid", 64, 12, 777), ("ppo", 128, 3, 2048),
                                         ("modular", 128, 3, 65536),
                                         ("modular", 128, 3, 70000),   # > 256 sample groups: workgroups accumulate a 2nd group
                                         ("resid", 128, 30, 2048),     # stacked-tank width: LDS map does not fit -> streamed 16-tile family
                                         ("modular", 128, 30, 2048),   # wide modular actor: the split net + dW pipeline
                                         ("modular", 128, 6, 3000),
                                         ("modular", 128, 10, 2048),   # first-layer gradients in matrix form (fan-in > 8)
                                         ("resid", 128, 10, 1500)])
def test_fused_gradients_match_autograd(kind, md, D, B):
    from pime_amd import ops
    act, cri = _make(kind, md, D, seed=B + md)
    L = max(3 * B, 5000)
    state, action, logprob, adv, r_sum = _data(L, D, act, seed=1)
    idx = torch.randint(L, (B,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    clip, lam = 0.2, 0.02
    want, s_sur, s_ent, s_cri, scale = _torch_grads(act, cri, state, action, logprob, adv, r_sum, idx, clip, lam)
    fused = ops.FusedPPOGrad(act, cri, B)
    fused.zero_grad()
    fused.loss_sums.zero_()
    got_scale = torch.zeros(1, device=DEV)
    fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, idx, clip, lam, got_scale)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got_scale.item(), scale.item(), rtol=3e-6)  # 1/(r_sum[idx].std()+1e-5), computed in-kernel
    got = {n: p.grad for n, p in list(act.named_parameters()) + [("cri." + k, v) for k, v in cri.named_parameters()]
           if p.requires_grad}
    assert set(got) == set(want)
    for name in want:
        w, g = want[name], got[name]
        # f32 sums over B samples in a different order: 3e-4 of the tensor's largest gradient entry
        tol = 3e-4 * float(w.abs().max()) + 1e-7
        err = float((w - g).abs().max())
        assert err <= tol, f"{name}: max |diff| {err:.3e} > {tol:.3e} (|grad|max {float(w.abs().max()):.3e})"
    sums = fused.loss_sums.tolist()
    np.testing.assert_allclose(sums[0], s_sur, rtol=2e-4, atol=1e-3 * B ** 0.5)
    np.testing.assert_allclose(sums[1], s_ent, rtol=2e-4, atol=1e-3 * B ** 0.5)
    np.testing.assert_allclose(sums[2], s_cri, rtol=2e-4)


def test_fused_gradients_are_reproducible_and_accumulate():
    """The per-workgroup gradient slabs are summed in a fixed order: two calls give bit-identical gradients; and the
    ABI contract is ACCUMULATE (include/pime_hip.h): a second call without zeroing doubles them."""
    from pime_amd import ops
    B = 8192
    act, cri = _make("modular", 128, 3, seed=5)
    state, action, logprob, adv, r_sum = _data(3 * B, 3, act, seed=3)
    idx = torch.randint(3 * B, (B,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    fused = ops.FusedPPOGrad(act, cri, B)
    scale = torch.zeros(1, device=DEV)

    def run(zero):
        if zero:
            fused.zero_grad()
        fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, idx, 0.2, 0.02, scale)
        torch.cuda.synchronize()
        return fused.flat_grad.clone()

    g1, g2 = run(True), run(True)
    assert torch.equal(g1, g2), "gradients differ between two identical calls"
    g3 = run(False)
    torch.testing.assert_close(g3, 2 * g1, rtol=1e-6, atol=1e-9)
    # PIME_PPO_OVERWRITE_GRADS: the same gradients whatever the buffers held; loss_sums[3] keeps a running sum of the scale
    fused.flat_grad.fill_(123.0)
    before = float(fused.loss_sums[3])
    fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, idx, 0.2, 0.02, scale, overwrite=True)
    torch.cuda.synchronize()
    assert torch.equal(fused.flat_grad, g1)
    np.testing.assert_allclose(float(fused.loss_sums[3]) - before, float(scale), rtol=1e-6)
    # index table + device-side row cursor (what the captured one-graph optimizer step uses): row 1 of the table == idx
    table = torch.randint(3 * B, (3, B), device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    table[1] = idx
    row = torch.ones(1, dtype=torch.int64, device=DEV)
    fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, table, 0.2, 0.02, scale, overwrite=True, index_row=row)
    torch.cuda.synchronize()
    assert torch.equal(fused.flat_grad, g1) and int(row) == 2


def test_split_pipeline_honours_overwrite_and_index_table():
    """The atomics-based pipeline that serves a modular actor on a wide state (LDS map of the fused kernel does not fit;
    plain nets go to the streamed 16-tile family instead): the same ABI options -- PIME_PPO_OVERWRITE_GRADS zeroes its
    targets first, index_row selects and advances the table row."""
    from pime_amd import ops
    B, D = 2048, 30
    act, cri = _make("modular", 128, D, seed=11)
    state, action, logprob, adv, r_sum = _data(3 * B, D, act, seed=5)
    idx = torch.randint(3 * B, (B,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))
    want, *_ = _torch_grads(act, cri, state, action, logprob, adv, r_sum, idx, 0.2, 0.02)
    fused = ops.FusedPPOGrad(act, cri, B)
    scale = torch.zeros(1, device=DEV)
    table = torch.zeros((2, B), dtype=torch.int64, device=DEV)
    table[1] = idx
    row = torch.ones(1, dtype=torch.int64, device=DEV)
    fused.flat_grad.fill_(7.0)
    fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, table, 0.2, 0.02, scale, overwrite=True, index_row=row)
    torch.cuda.synchronize()
    assert int(row) == 2
    got = {n: p.grad for n, p in list(act.named_parameters()) + [("cri." + k, v) for k, v in cri.named_parameters()]
           if p.requires_grad}
    for name in want:
        tol = 3e-4 * float(want[name].abs().max()) + 1e-7
        assert float((want[name] - got[name]).abs().max()) <= tol, name


def test_fused_update_net_matches_torch_update():
    """Whole update_net: the fused path and the torch-autograd path, same start, same minibatch indices."""
    from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from pime_amd.elegantrl.replay import TrajectoryBuffer
    T, N, D = 50, 512, 3
    idx_gen = torch.Generator(device="cpu").manual_seed(5)
    idx_all = [torch.randint(T * N, (4096,), generator=idx_gen) for _ in range(64)]
    out = []
    for fused in (False, True):
        torch.manual_seed(0)
        ag = AgentResidualIntegratorModularPPO(device=DEV)
        ag.lambda_gae_adv = 0.99
        ag.init(128, D, 1, 1)
        ag.init_residual({"init_K": np.array([[-0.02], [0.02], [0.035]])})
        with torch.no_grad():
            ag.act.net[-1].weight.normal_(0, 0.05)
        ag.weights_changed()
        ag.use_fused_update = fused
        ag.index_hook = lambda step, L, B: idx_all[step]
        buf = TrajectoryBuffer(T, N, D, 1, DEV)
        g = torch.Generator(device="cpu").manual_seed(9)
        buf.state[:T] = (torch.randn(T, N, D, generator=g) * torch.tensor([3., 3., 8.]) + torch.tensor([7., 7., 0.])).to(DEV)
        buf.reward[:] = -(torch.rand(T, N, generator=g) * 20).to(DEV)
        buf.mask[:] = 0.99
        buf.mask[-1] = 0
        buf.noise[:] = torch.randn(T, N, 1, generator=g).to(DEV)
        with torch.no_grad():
            flat = buf.state[:T].reshape(-1, D)
            buf.action[:] = (ag.act.mean(flat) + buf.noise.reshape(-1, 1) * ag.act.a_std_log.exp()).reshape(T, N, 1)
        buf.length = T
        oa, oc = ag.update_net(buf, T * N, 4096, 1.0)   # 6 optimizer steps
        out.append(({k: v.detach().clone() for k, v in list(ag.act.state_dict().items()) + list(ag.cri.state_dict().items())}, oa, oc))
    (w0, a0, c0), (w1, a1, c1) = out
    for k in w0:
        np.testing.assert_allclose(w1[k].cpu().numpy(), w0[k].cpu().numpy(), rtol=0, atol=3e-5, err_msg=k)  # lr 1e-4 x 6 Adam steps
    np.testing.assert_allclose(a1, a0, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(c1, c0, rtol=1e-3)


@pytest.mark.parametrize("env_name,algo", [("PH_V35", "ResidualIntegratorModularPPO"), ("PH_V35", "ResidualPPO"),
                                           ("PH_V35", "PPO"), ("WT_INTEGRATOR", "ResidualIntegratorModularPPO"),
                                           ("WT_INTEGRATOR", "ResidualPPO")])
def test_fused_rollout_matches_stepwise_rollout(env_name, algo):
    """pime_rollout (one launch per episode) against the per-step launch sequence with the SAME exploration noise:
    the step-wise agent replays the noise the fused kernel drew (noise_hook), the envs share seed and Philox streams
    (resets AND water-tank process noise), so states / actions / rewards / done flags must agree step for step over
    two episodes (auto-reset in between).  The env arithmetic is literally the same device functions
    (csrc/env_device.hpp); the policy mean comes from two MFMA chains with different summation orders (16-lane tiles in the
    fused kernel, 32-lane tiles in the forward kernel), i.e. ulps -> for pH 'within one LUT cell' on rare lanes, which are dropped
    once they diverge."""
    from pime_amd import gym_control
    from pime_amd.elegantrl.run import make_buffer
    from pime_amd.utils import MODELS
    N = 1024
    is_ph = env_name == "PH_V35"
    env_id = getattr(gym_control, env_name)
    kw = {} if is_ph else dict(reward_type="distance", max_step=40)
    bufs = []
    for fused in (True, False):
        env = gym_control.make_vec(env_id, N, device=DEV, seed=11, **kw)
        T = env.max_step
        torch.manual_seed(0)
        ag = MODELS[algo.lower()](device=DEV)
        if "modular" in algo.lower():
            ag.init(128, env.state_dim, 1, 1)
        else:
            ag.init(128, env.state_dim, 1)
        if "residual" in algo.lower():
            ag.init_residual({"init_K": env.K.reshape(-1, 1)})
        with torch.no_grad():
            ag.act.net[-1].weight.normal_(0, 0.1)
        ag.weights_changed()
        ag.use_fused_rollout = fused
        buf = make_buffer(ag, env, 2 * N * T)
        if not fused:
            ref_noise = bufs[0].noise.clone()
            ag.noise_hook = lambda t, shape: ref_noise[t].reshape(shape)
        else:
            assert ag._fused_rollout_ok(env), "fused rollout path not available"
        steps = ag.explore_env(env, buf, 2 * N * T, 1.0, 0.99)
        assert steps == 2 * N * T
        bufs.append(buf)
        env.close()
    f, s = bufs
    noise = f.noise[:2 * T].cpu().numpy()
    assert abs(noise.mean()) < 0.02 and abs(noise.std() - 1.0) < 0.02      # N(0,1) exploration noise
    np.testing.assert_array_equal(f.done[:2 * T].cpu().numpy(), s.done[:2 * T].cpu().numpy())
    assert f.done[T - 1].all() and f.done[2 * T - 1].all() and not f.done[:T - 1].any()
    np.testing.assert_array_equal(f.state[0].cpu().numpy(), s.state[0].cpu().numpy())
    alive = np.ones(N, dtype=bool)
    # pH: each path's recorded actions are also replayed through its own OraclePH (as tests/rollout_replay.py does), so that the
    # episode's LAST step -- whose y is not stored: state[t + 1] is the next episode's first observation -- is held to the oracle's
    # reward on EVERY lane of BOTH paths instead of to the other path within an allowance
    refs = None
    if is_ph:
        import oracle
        refs = [oracle.OraclePH(N, oracle.ph_table(), seed=11) for _ in range(2)]
        for r_ in refs:
            r_.reset()
        priorK = ag._rollout_priorK()
    for t in range(2 * T):
        if t == T:
            alive[:] = True   # a new episode starts from the same Philox reset draws on every lane
        fa, sa = f.action[t, :, 0].cpu().numpy(), s.action[t, :, 0].cpu().numpy()
        fs, ss = f.state[t + 1].cpu().numpy(), s.state[t + 1].cpu().numpy()
        np.testing.assert_allclose(fa[alive], sa[alive], rtol=1e-4, atol=1e-5)
        want_rew = None
        if is_ph:
            want_rew = []
            for buf_, ref_, act_ in ((f, refs[0], fa), (s, refs[1], sa)):
                env_act = oracle.residual_action(act_, buf_.state[t].cpu().numpy(), priorK)
                want_rew.append(ref_.step(env_act, auto_reset=True)[2])
        if is_ph:   # a lane whose action differs in the last bit may read the neighbouring titration cell at THIS step: its reward
            dy = np.abs(fs[:, 0] - ss[:, 0])   # and next state differ from here on, so it leaves the comparison before they are checked
            if t not in (T - 1, 2 * T - 1):
                assert dy[alive].max() <= 0.0297
            alive &= dy <= 1e-5
        fr_all, sr_all = f.reward[t].cpu().numpy(), s.reward[t].cpu().numpy()
        fr, sr = fr_all[alive], sr_all[alive]
        if is_ph and t in (T - 1, 2 * T - 1):
            # the episode's last step: every lane of either path against the oracle's reward for that path's own action
            for got, want in ((fr_all, want_rew[0]), (sr_all, want_rew[1])):
                np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5, err_msg=f"last-step reward, step {t}")
        else:
            np.testing.assert_allclose(fr, sr, rtol=3e-4, atol=3e-4)
        np.testing.assert_allclose(fs[alive], ss[alive], rtol=1e-4, atol=1e-4)
    assert alive.mean() > 0.95


@pytest.mark.parametrize("kind,md,D", [("modular", 128, 3), ("resid", 64, 4), ("resid", 256, 30), ("resid", 128, 30)])
def test_adam_fused_into_the_slab_reduction_equals_the_separate_step(kind, md, D):
    """pime_ppo_minibatch_step (gradients + Adam in the slab reduction) against pime_ppo_minibatch_grad + pime_adam_step:
    the same gradient element feeds the same update arithmetic, so parameters, moments and step counter are bit-equal."""
    from pime_amd import ops
    B = 4096
    outs = []
    for fuse in (False, True):
        act, cri = _make(kind, md, D, seed=7)
        state, action, logprob, adv, r_sum = _data(3 * B, D, act, seed=3)
        fused = ops.FusedPPOGrad(act, cri, B)
        adam = fused.make_optimizer(1e-3)
        scale = torch.zeros(1, device=DEV)
        for step in range(3):
            idx = torch.randint(3 * B, (B,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(10 + step))
            fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, idx, 0.2, 0.02, scale, overwrite=True,
                  adam=adam if fuse else None)
            if not fuse:
                adam.step()
            fused.repack()
        torch.cuda.synchronize()
        outs.append((fused.flat_param.clone(), adam.exp_avg.clone(), adam.exp_avg_sq.clone(), adam.step_count.clone(),
                     fused.flat_grad.clone()))
    for a, b, name in zip(outs[0], outs[1], ("param", "exp_avg", "exp_avg_sq", "step", "grad")):
        assert torch.equal(a, b), f"{name} differs between the fused and the separate optimizer step"
    assert float(outs[1][3][0]) == 3.0 and float(outs[1][3][1]) == 0.0


def test_split_pipeline_refuses_the_fused_step():
    from pime_amd import native, ops
    B, D = 1024, 30
    act, cri = _make("modular", 128, D, seed=11)      # wide modular actor -> split pipeline
    state, action, logprob, adv, r_sum = _data(3 * B, D, act, seed=5)
    fused = ops.FusedPPOGrad(act, cri, B)
    adam = fused.make_optimizer(1e-3)
    idx = torch.randint(3 * B, (B,), device=DEV)
    with pytest.raises(native.PimeError, match="split pipeline"):
        fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, idx, 0.2, 0.02, torch.zeros(1, device=DEV),
              overwrite=True, adam=adam)


@pytest.mark.parametrize("kind,md,D", [("modular", 128, 3), ("resid", 128, 3), ("resid", 64, 4), ("resid", 256, 3), ("resid", 128, 30)])
def test_fused_step_keeps_the_packed_images_current(kind, md, D):
    """pime_ppo_image_map + pime_adam.image_map: the launch that applies Adam writes every new parameter value into the nets'
    forward / transposed images, so the re-pack launch after an optimizer step can go.  After three fused steps the images must
    equal, bit for bit, what pime_ppo_repack lays out from the updated parameters -- for the LDS-resident family (64 / 128), the
    16-tile family (256, and 128 on the 30-float observation) and the modular actor."""
    from pime_amd import ops
    act, cri = _make(kind, md, D, seed=5)
    B = 2048
    L = 3 * B
    state, action, logprob, adv, r_sum = _data(L, D, act, seed=3)
    fused = ops.FusedPPOGrad(act, cri, B)
    opt = fused.make_optimizer(3e-4)
    assert fused.images_follow_step, f"the library could not derive the image map: {fused.image_map_error}"
    m = fused.image_map().view(-1, 2)
    assert int((m[:, 0] >= 0).sum()) >= fused.flat_param.numel() - 2   # every net parameter sits in its forward image (a_std_log does not)
    scale = torch.zeros(1, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(11)
    for _ in range(3):
        idx = torch.randint(L, (B,), device=DEV, generator=g)
        fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, idx, 0.2, 0.02, scale, overwrite=True, adam=opt)
    torch.cuda.synchronize()
    got = [(n["img_fwd"].clone(), n["img_bwd"].clone()) for n in fused.nets]
    fused.repack()
    torch.cuda.synchronize()
    for (gf, gb), n in zip(got, fused.nets):
        assert torch.equal(gf, n["img_fwd"]), "forward image drifted from the parameters"
        assert torch.equal(gb, n["img_bwd"]), "transposed image drifted from the parameters"


@pytest.mark.parametrize("kind,md,D", [("modular", 128, 3), ("resid", 256, 3)])
def test_separate_adam_launch_keeps_the_packed_images_current(kind, md, D):
    """The data-parallel form (gradients, all-reduce, THEN Adam): pime_adam_step_images updates parameters and packed images in
    the one Adam launch; the parameters must equal pime_adam_step's bit for bit and the images a re-pack's."""
    from pime_amd import ops
    act, cri = _make(kind, md, D, seed=6)
    act2, cri2 = _make(kind, md, D, seed=6)
    B = 1024
    L = 3 * B
    state, action, logprob, adv, r_sum = _data(L, D, act, seed=4)
    fa, fb = ops.FusedPPOGrad(act, cri, B), ops.FusedPPOGrad(act2, cri2, B)
    oa, ob = fa.make_optimizer(3e-4), fb.make_optimizer(3e-4)
    scale = torch.zeros(1, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(12)
    for _ in range(3):
        idx = torch.randint(L, (B,), device=DEV, generator=g)
        for f, o, images in ((fa, oa, fa), (fb, ob, None)):
            f(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, idx, 0.2, 0.02, scale, overwrite=True)
            o.step(images=images)
            if images is None:
                f.repack()
    torch.cuda.synchronize()
    assert torch.equal(fa.flat_param, fb.flat_param)
    for na, nb in zip(fa.nets, fb.nets):
        assert torch.equal(na["img_fwd"], nb["img_fwd"]) and torch.equal(na["img_bwd"], nb["img_bwd"])


def test_full_size_gradient_is_the_sum_over_a_partition_of_the_minibatch():
    """A size-independent property at the bench's full minibatch (65 536 of 819 200 transitions, modular actor + critic, width
    128): the losses are means over the minibatch, so the gradient of the whole minibatch is the mean of the gradients of its two
    halves (the critic's after undoing each call's 1/(std+1e-5) scale).  Checked to 2e-5 of each tensor's largest entry: the two
    sides sum 65 536 float32 terms in different orders."""
    from pime_amd import ops
    act, cri = _make("modular", 128, 3, seed=9)
    L, B = 819200, 65536
    state, action, logprob, adv, r_sum = _data(L, 3, act, seed=8)
    fused = ops.FusedPPOGrad(act, cri, B)
    idx = torch.randint(L, (B,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(21))
    scale = torch.zeros(1, device=DEV)
    n_act = sum(p.numel() for p in act.parameters() if p.requires_grad)

    def grad_of(ix):
        fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, ix.contiguous(), 0.2, 0.02, scale, overwrite=True)
        torch.cuda.synchronize()
        g = fused.flat_grad.clone()
        g[n_act:] /= scale.item()          # the critic's gradients without the per-call scale (agent.py:652)
        return g
    whole, h1, h2 = grad_of(idx), grad_of(idx[:B // 2]), grad_of(idx[B // 2:])
    mean = 0.5 * (h1 + h2)
    off = 0
    for p in fused.params:
        n = p.numel()
        w, m = whole[off:off + n], mean[off:off + n]
        tol = 2e-5 * float(w.abs().max()) + 1e-9
        assert float((w - m).abs().max()) <= tol, f"parameter at flat offset {off}: {float((w - m).abs().max()):.3e} > {tol:.3e}"
        off += n
