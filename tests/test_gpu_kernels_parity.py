"""GPU parity of the trajectory/MLP kernels (GAE scan, fused MFMA forwards) against the CPU oracle and the golden
vectors, through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sd(g, tag):
    return {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + ".")}


@pytest.mark.parametrize("T,N", [(50, 8), (50, 16384), (200, 4096), (7, 129)])
@pytest.mark.parametrize("use_gae", [True, False])
def test_gae_scan(T, N, use_gae):
    from pime_amd import ops
    rng = np.random.RandomState(T * 1000 + N)
    rew = (rng.standard_normal((T, N)) * 3 - 2).astype(np.float32)
    val = rng.standard_normal((T, N)).astype(np.float32)
    mask = np.full((T, N), 0.99, dtype=np.float32)
    mask[-1] = 0
    mask[rng.randint(0, T, 5), rng.randint(0, N, 5)] = 0
    r_sum, adv = ops.gae_scan(torch.as_tensor(rew, device=DEV), torch.as_tensor(mask, device=DEV),
                              torch.as_tensor(val, device=DEV), 0.97, use_gae)
    w_r, w_a = oracle.gae(rew, mask, val, 0.97, use_gae)
    # same float32 operations in the same order, no contraction: bitwise
    np.testing.assert_array_equal(r_sum.cpu().numpy(), w_r)
    np.testing.assert_array_equal(adv.cpu().numpy(), w_a)


def test_gae_golden():
    from pime_amd import ops
    g = load_golden("gae.npz")
    t = lambda k: torch.as_tensor(np.ascontiguousarray(g[k].T), device=DEV)  # noqa: E731  golden is [lane, T]
    r_sum, adv = ops.gae_scan(t("reward"), t("mask"), t("value"), 0.97, True)
    np.testing.assert_allclose(r_sum.cpu().numpy().T, g["r_sum_0.97"], rtol=2e-6, atol=2e-6)
    a = adv.double()
    a = ((a - a.mean()) / (a.std() + 1e-5)).cpu().numpy().T
    np.testing.assert_allclose(a, g["adv_0.97"], rtol=2e-5, atol=2e-5)


def _torch_sd(sd):
    return {k: torch.as_tensor(v, device=DEV) for k, v in sd.items()}


@pytest.mark.parametrize("M", [64, 1, 33, 4096, 100003])
def test_mlp_forward_golden_weights(M):
    """Fused MFMA forwards with the reference-initialised weights from nets.npz against the oracle's
    double-accumulated forward.  f32 MFMA = k-ordered f32 FMA chain: 1e-5 relative on O(1) outputs."""
    from pime_amd import ops
    g = load_golden("nets.npz")
    rng = np.random.RandomState(M)
    x3 = g["x3"] if M == 64 else (rng.standard_normal((M, 3)) * [4, 4, 10] + [7, 7, 0]).astype(np.float32)
    xt = torch.as_tensor(x3, device=DEV)
    sd = _sd(g, "critic3")
    v = ops.PackedMLP.from_state_dict("critic", _torch_sd(sd), state_dim=3)(xt)
    np.testing.assert_allclose(v.cpu().numpy(), oracle.critic_forward(x3, sd)[:, 0], rtol=2e-5, atol=2e-5)
    sd = _sd(g, "modular3")
    a = ops.PackedMLP.from_state_dict("modular_actor", _torch_sd(sd), state_dim=3, integrator_dim=1)(xt)
    np.testing.assert_allclose(a.cpu().numpy(), oracle.modular_actor_mean(x3, sd)[:, 0], rtol=2e-5, atol=2e-5)
    if M == 64:
        np.testing.assert_allclose(v.cpu().numpy(), g["critic3:forward"][:, 0], rtol=2e-5, atol=2e-5)
        std = np.exp(sd["a_std_log"])
        np.testing.assert_allclose(a.cpu().numpy()[:, None] + g["eps"] * std, g["modular3:action"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("kind,D,Di,md", [("critic", 4, 0, 64), ("critic", 30, 0, 128), ("plain_actor", 3, 0, 128),
                                          ("plain_actor", 12, 0, 64), ("modular_actor", 4, 1, 64),
                                          ("modular_actor", 4, 1, 128), ("modular_actor", 3, 1, 128)])
def test_mlp_forward_random_weights(kind, D, Di, md):
    from pime_amd import ops
    rng = np.random.RandomState(D * 100 + md)
    M = 777
    x = rng.standard_normal((M, D)).astype(np.float32)

    def lin(o, i, scale=1.0):
        return (rng.standard_normal((o, i)) * scale / np.sqrt(i)).astype(np.float32), (rng.standard_normal(o) * 0.3).astype(np.float32)
    sd = {}
    if kind == "modular_actor":
        shapes = [("other_net.0", md, D - Di), ("other_net.2", md // 2, md), ("integrator_net.0", md, Di),
                  ("integrator_net.2", md // 2, md), ("net.0", md, md), ("net.2", 1, md)]
    else:
        shapes = [("net.0", md, D), ("net.2", md, md), ("net.4", md, md), ("net.6", 1, md)]
    for name, o, i in shapes:
        sd[name + ".weight"], sd[name + ".bias"] = lin(o, i, 1.5)
    out = ops.PackedMLP.from_state_dict(kind, _torch_sd(sd), state_dim=D, integrator_dim=Di)(torch.as_tensor(x, device=DEV))
    want = dict(critic=oracle.critic_forward, plain_actor=oracle.plain_actor_mean)[kind](x, sd)[:, 0] if kind != "modular_actor" \
        else oracle.modular_actor_mean(x, sd, Di)[:, 0]
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=3e-5, atol=3e-5)


def test_mlp_forward_matches_torch_fp32():
    """Against a plain PyTorch fp32 forward of the same nets on the GPU (rocBLAS)."""
    from pime_amd import ops
    from pime_amd.elegantrl.net import CriticAdv
    from pime_amd.elegantrl.net_residual import ActorResidualIntegratorModularPPO
    torch.manual_seed(0)
    cri = CriticAdv(3, 128).to(DEV)
    act = ActorResidualIntegratorModularPPO(128, 3, 1, 1).to(DEV)
    with torch.no_grad():
        act.net[-1].weight.mul_(8)
    x = torch.randn(50000, 3, device=DEV) * torch.tensor([4., 4., 10.], device=DEV)
    with torch.no_grad():
        np.testing.assert_allclose(ops.PackedMLP.from_module(cri)(x).cpu().numpy(), cri(x)[:, 0].cpu().numpy(), rtol=3e-5, atol=3e-5)
        np.testing.assert_allclose(ops.PackedMLP.from_module(act)(x).cpu().numpy(), act.mean(x)[:, 0].cpu().numpy(), rtol=3e-5, atol=3e-5)


def test_flat_adam_matches_torch_adam():
    """pime_adam_step (one launch, device-side step counter advanced by the last workgroup) against torch.optim.Adam
    (agent.py:565-566: lr only, default betas / eps, no weight decay) over several steps, eager and from a HIP graph."""
    from pime_amd import ops
    torch.manual_seed(3)
    n = 67460
    p0 = torch.randn(n, device=DEV)
    grads = [torch.randn(n, device=DEV) * (0.1 + k) for k in range(6)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-4)
    mine = p0.clone()
    g = torch.zeros(n, device=DEV)
    adam = ops.FlatAdam(mine, g, 1e-4)
    for k in range(3):
        ref.grad = grads[k].clone(); opt.step()
        g.copy_(grads[k]); adam.step()
    torch.cuda.synchronize()
    assert float(adam.step_count[0]) == 3.0
    torch.testing.assert_close(mine, ref.detach(), rtol=2e-6, atol=2e-7)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        adam.step()
    # (capture does not execute) three replays = steps 4..6
    for k in range(3, 6):
        ref.grad = grads[k].clone(); opt.step()
        g.copy_(grads[k]); graph.replay()
    torch.cuda.synchronize()
    assert float(adam.step_count[0]) == 6.0
    torch.testing.assert_close(mine, ref.detach(), rtol=2e-6, atol=2e-7)
