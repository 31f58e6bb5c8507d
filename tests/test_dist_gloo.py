"""Data-parallel path on CPU: world_size 2 over gloo (the GPU path uses the same code over RCCL).
Two ranks, each with its own slice of env lanes (different Philox lane offsets) and exploration stream; after
explore + update the replicas must be bit-identical, the flat-gradient all-reduce must equal the mean of the local
gradients, and the advantage normalisation must use the moments of the UNION of both slices (agent.py:707)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    from oracle.cpu_stack import OracleBackend, OracleVecEnv
    from pime_amd import dist as pdist
    from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from pime_amd.elegantrl.replay import TrajectoryBuffer
    dp = pdist.init_from_env(backend="gloo", device="cpu")
    assert dp is not None and dp.world == world and dp.rank == rank
    n = 32
    env = OracleVecEnv("ph", n, seed=5, env_offset=dp.lane_offset(n))
    torch.manual_seed(100 + rank)             # replicas start DIFFERENT on purpose ...
    agent = AgentResidualIntegratorModularPPO(backend=OracleBackend(), device="cpu")
    agent.lambda_gae_adv = 0.99
    agent.init(32, 3, 1, 1)
    agent.init_residual({"init_K": env.K.reshape(-1, 1)})
    with torch.no_grad():
        agent.act.net[-1].weight.normal_(0, 0.05)
    agent.dp = dp
    dp.broadcast_module(agent.act, agent.cri)  # ... and are made equal by the rank-0 broadcast
    torch.manual_seed(1000 + rank)             # exploration / minibatch streams differ per rank
    buf = TrajectoryBuffer(50, n, 3, 1, "cpu")
    steps = agent.explore_env(env, buf, n * 50, 1.0, 0.99)

    # (1) gradient averaging: local grads -> all-reduce -> compare with an all_gather'ed mean
    params = [p for g in agent.optimizer.param_groups for p in g["params"]]
    for p in params:
        p.grad = torch.full_like(p, float(rank + 1)) * torch.arange(p.numel(), dtype=torch.float32).view_as(p)
    local = torch.cat([p.grad.reshape(-1) for p in params]).clone()
    agent.dp.average_gradients(params)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    torch.distributed.all_gather(gathered, local)
    avg_ok = torch.allclose(torch.cat([p.grad.reshape(-1) for p in params]), torch.stack(gathered).mean(0))

    # (2) advantage normalisation over the union of both slices
    adv_local = torch.randn(n * 50, generator=torch.Generator().manual_seed(7 + rank)) * (rank + 1) + rank
    normed = agent._normalise_advantage(adv_local)
    all_adv = [torch.zeros_like(adv_local) for _ in range(world)]
    torch.distributed.all_gather(all_adv, adv_local)
    union = torch.cat(all_adv)
    want = (adv_local - union.mean()) / (union.std() + 1e-5)
    norm_ok = torch.allclose(normed, want, rtol=1e-5, atol=1e-5)

    # (3) a real update keeps the replicas identical
    agent.update_net(buf, n * 50, 256, 2)
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.act.parameters()) + list(agent.cri.parameters())])
    torch.save({"flat": flat, "avg_ok": bool(avg_ok), "norm_ok": bool(norm_ok), "steps": steps,
                "obs0": buf.state[0].clone()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dp.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo(tmp_path):
    import oracle
    oracle.build()  # compile the oracle once, before the ranks race for it
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{k}.pt"), weights_only=True) for k in range(world)]
    assert all(x["avg_ok"] for x in r), "flat-gradient all-reduce != mean of the local gradients"
    assert all(x["norm_ok"] for x in r), "advantage normalisation is not over the union of the ranks' buffers"
    assert r[0]["steps"] == r[1]["steps"] == 32 * 50
    assert torch.equal(r[0]["flat"], r[1]["flat"]), "replicas diverged"
    assert not torch.equal(r[0]["obs0"], r[1]["obs0"]), "ranks simulated the same env lanes (lane offset ignored)"


def _train_worker(rank, world, port, out_dir):
    """train_and_evaluate under data parallelism with a target_return only rank 0 can reach: rank 0 alone would leave the
    loop after the first evaluation and rank 1 would block forever in its next all-reduce (the loop-exit flag is now a
    MAX all-reduce); torch is re-seeded per rank after the weight broadcast; only rank 0 writes checkpoints."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from oracle.cpu_stack import OracleBackend, OracleVecEnv
    from pime_amd import dist as pdist
    from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from pime_amd.elegantrl.run import Arguments, train_and_evaluate
    dp = pdist.init_from_env(backend="gloo", device="cpu")
    n = 16
    env = OracleVecEnv("ph", n, seed=5, env_offset=dp.lane_offset(n))
    env.env_name = "ph-oracle"
    env.target_return = -1e9 if rank == 0 else 1e9      # reachable on rank 0 only
    args = Arguments(if_on_policy=True)
    args.agent = AgentResidualIntegratorModularPPO(backend=OracleBackend(), device="cpu")
    args.agent.dp = dp
    args.env = env
    args.cwd = os.path.join(out_dir, "run")              # the same directory on every rank, as train.py arranges
    args.if_remove = False
    args.net_dim, args.batch_size, args.repeat_times, args.target_step, args.max_memo = 32, 128, 1, n * 50, n * 50
    args.break_step = 10 * n * 50
    args.eval_gap, args.eval_times1, args.eval_times2 = 1, n, n
    args.num_threads = 1
    args.random_seed = 3
    args.residual_kwargs = {"init_K": env.K.reshape(-1, 1)}
    args.Modular_kwargs = {"integrator_dim": 1}
    args.if_residual = True
    args.fix_K = True
    agent, buf = train_and_evaluate(args)
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.act.parameters()) + list(agent.cri.parameters())])
    torch.save({"flat": flat, "noise": buf.noise[:4].clone(), "length": buf.length, "seed": torch.initial_seed()},
               os.path.join(out_dir, f"train_rank{rank}.pt"))
    dp.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_train_loop_exits_together(tmp_path):
    import oracle
    oracle.build()
    world, port = 2, _free_port()
    mp.spawn(_train_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"train_rank{k}.pt"), weights_only=True) for k in range(world)]
    assert torch.equal(r[0]["flat"], r[1]["flat"]), "replicas diverged"
    assert r[0]["seed"] == 3 and r[1]["seed"] == 4, "torch was not re-seeded with random_seed + rank"
    # the goal is 'reached' (rank 0) at the evaluation before the loop: no training iteration runs on EITHER rank
    assert r[0]["length"] == r[1]["length"] == 0
    run = os.path.join(tmp_path, "run")
    assert os.path.exists(os.path.join(run, "actor.pth")) and os.path.exists(os.path.join(run, "init", "critic.pth"))


# ---- the critic scale of agent.py:652 on the UNION minibatch -------------------------------------------------------------------------
def _fill(buf, gen):
    """Synthetic trajectory rows (no env needed: the update only sees the buffer)."""
    T, n = buf.horizon, buf.num_envs
    buf.state[:T + 1] = torch.randn(T + 1, n, 3, generator=gen) * torch.tensor([2.0, 2.0, 5.0]) + torch.tensor([7.0, 7.0, 0.0])
    buf.reward[:T] = -torch.rand(T, n, generator=gen) * 4
    buf.mask[:T] = torch.where(torch.rand(T, n, generator=gen) < 0.05, 0.0, 0.99)
    buf.noise[:T] = torch.randn(T, n, 1, generator=gen)
    buf.action[:T] = torch.randn(T, n, 1, generator=gen) * 0.6
    buf.length = T


def _union_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from oracle.cpu_stack import OracleBackend
    from pime_amd import dist as pdist
    from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from pime_amd.elegantrl.replay import TrajectoryBuffer
    dp = pdist.init_from_env(backend="gloo", device="cpu")
    T, n, B, n_steps = 20, 16, 64, 3

    def make_agent():
        torch.manual_seed(42)                      # the same initial weights everywhere
        ag = AgentResidualIntegratorModularPPO(backend=OracleBackend(), device="cpu")
        ag.lambda_gae_adv = 0.99
        ag.init(32, 3, 1, 1)
        ag.init_residual({"init_K": np.array([[-0.02], [0.02], [0.035]])})
        with torch.no_grad():
            ag.act.net[-1].weight.normal_(0, 0.05)
        return ag

    agent = make_agent()
    agent.dp = dp
    buf = TrajectoryBuffer(T, n, 3, 1, "cpu")
    _fill(buf, torch.Generator().manual_seed(500 + rank))          # every rank holds different lanes ...
    idx = torch.randint(T * n, (n_steps, B), generator=torch.Generator().manual_seed(900 + rank))   # ... and draws its own rows
    agent.index_hook = lambda step, buf_len, batch: idx[step]
    agent.update_net(buf, T * n, B, n_steps * B / (T * n))
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.act.parameters()) + list(agent.cri.parameters())])
    # rank 0 gathers everybody's rows and repeats the update ALONE on the union buffer with the concatenated minibatches
    rows = {k: getattr(buf, k)[:T + (1 if k == "state" else 0)].clone() for k in ("state", "reward", "mask", "noise", "action")}
    gathered = {}
    for k, v in rows.items():
        parts = [torch.zeros_like(v) for _ in range(world)]
        torch.distributed.all_gather(parts, v.contiguous())
        gathered[k] = torch.cat(parts, dim=1)                       # lanes of rank r at [r n, (r + 1) n)
    all_idx = [torch.zeros_like(idx) for _ in range(world)]
    torch.distributed.all_gather(all_idx, idx)
    out = {"flat": flat}
    if rank == 0:
        solo = make_agent()
        big = TrajectoryBuffer(T, n * world, 3, 1, "cpu")
        for k, v in gathered.items():
            getattr(big, k)[:v.shape[0]] = v
        big.length = T
        N = n * world

        def union_rows(step, buf_len, batch):       # local flat row t n + l of rank r = union row t N + r n + l
            parts = []
            for r in range(world):
                t, l = all_idx[r][step] // n, all_idx[r][step] % n
                parts.append(t * N + r * n + l)
            return torch.cat(parts)
        solo.index_hook = union_rows
        solo.update_net(big, T * N, B * world, n_steps * B * world / (T * N))
        out["solo"] = torch.cat([p.detach().reshape(-1) for p in list(solo.act.parameters()) + list(solo.cri.parameters())])
    torch.save(out, os.path.join(out_dir, f"union_rank{rank}.pt"))
    dp.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_data_parallel_update_equals_one_rank_on_the_union_minibatch(tmp_path, world):
    """/root/reference/elegantrl/agent.py:652 divides the critic loss by the std of the MINIBATCH's targets; on G ranks the
    minibatch is the union of the ranks' minibatches.  Every rank back-propagates the unscaled critic loss, the one flat all-reduce
    of the step carries (sum r, sum r^2, count) behind the gradients, and the scale of the union is applied to the averaged critic
    gradient: after three optimizer steps the replicas are bit-identical AND equal (2e-6) to ONE rank updating on the concatenated
    buffer with the concatenated minibatches (advantage normalisation over the union included)."""
    import oracle
    oracle.build()
    port = _free_port()
    mp.spawn(_union_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"union_rank{k}.pt"), weights_only=True) for k in range(world)]
    for k in range(1, world):
        assert torch.equal(r[0]["flat"], r[k]["flat"]), "replicas diverged"
    np.testing.assert_allclose(r[0]["flat"].numpy(), r[0]["solo"].numpy(), rtol=0, atol=2e-6)


class _ListBuffer:
    """A replay buffer that hands out prepared minibatches: (reward, mask, action, state, next_state) per optimizer step."""

    def __init__(self, batches):
        self.batches, self.i = batches, 0

    def update_now_len_before_sample(self):
        pass

    def sample_batch(self, batch_size):
        b = self.batches[self.i]
        self.i += 1
        assert b[0].shape[0] == batch_size
        return b


def _td3_union_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from oracle.cpu_stack import OracleBackend
    from pime_amd import dist as pdist
    from pime_amd.elegantrl.agent import AgentTD3
    dp = pdist.init_from_env(backend="gloo", device="cpu")
    B, n_steps, D = 48, 4, 4

    def make_agent():
        torch.manual_seed(7)
        ag = AgentTD3(backend=OracleBackend(), device="cpu")
        ag.init(32, D, 1)
        ag.policy_noise = 0.0      # no smoothing-noise draws: the union step is then a deterministic function of the minibatches
        return ag

    def batches(r):
        g = torch.Generator().manual_seed(300 + r)
        out = []
        for _ in range(n_steps):
            s = torch.randn(B, D, generator=g) * 2
            out.append((-torch.rand(B, 1, generator=g) * 3, torch.where(torch.rand(B, 1, generator=g) < 0.1, 0.0, 0.99),
                        torch.rand(B, 1, generator=g) * 2 - 1, s, s + 0.1 * torch.randn(B, D, generator=g)))
        return out

    agent = make_agent()
    agent.dp = dp
    mine = batches(rank)
    agent.update_net(_ListBuffer(mine), n_steps, B, 1.0)   # a non-vector buffer: int(target_step * repeat_times) optimizer steps
    nets = lambda a: torch.cat([p.detach().reshape(-1) for m in (a.act, a.cri, a.act_target, a.cri_target) for p in m.parameters()])
    out = {"flat": nets(agent)}
    if rank == 0:
        solo = make_agent()
        every = [batches(r) for r in range(world)]
        union = [tuple(torch.cat([every[r][k][j] for r in range(world)]) for j in range(5)) for k in range(n_steps)]
        solo.update_net(_ListBuffer(union), n_steps, B * world, 1.0)
        out["solo"] = nets(solo)
    torch.save(out, os.path.join(out_dir, f"td3_union_rank{rank}.pt"))
    dp.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_td3_data_parallel_update_equals_one_rank_on_the_union_minibatch(tmp_path):
    """AgentTD3.update_net under data parallelism (the reference has no collective): both backward passes of every optimizer step
    (agent.py:314-331) are followed by an all-reduce (mean) of that net's gradients.  Two ranks with different minibatches end four
    steps -- two of them with the delayed soft update -- bit-identical to each other and equal (2e-6) to ONE rank stepping on the
    concatenated minibatches: online nets and targets."""
    world = 2
    port = _free_port()
    mp.spawn(_td3_union_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"td3_union_rank{k}.pt"), weights_only=True) for k in range(world)]
    assert torch.equal(r[0]["flat"], r[1]["flat"]), "replicas diverged"
    np.testing.assert_allclose(r[0]["flat"].numpy(), r[0]["solo"].numpy(), rtol=0, atol=2e-6)
