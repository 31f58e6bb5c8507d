"""BASELINE.json config 4 on ONE GPU: "pH-process env, 131 072 instances sharded across 8 x MI355X".

Rank g of the 8-GPU job owns the global lanes [g * 16384, (g + 1) * 16384) -- its env is
`make_vec(PH_V35, 16384, env_offset=g * 16384, seed=s)` (bench.py:build_stack, train.py:make_env) -- and
nothing in a rollout crosses lanes, so the sharding is checked here without 8 GPUs: the eight rank slices are built one after
the other on cuda:0, each runs one fused-rollout episode under the same policy and is replayed through `OraclePH` with the
same lane offset (exploration noise bit-equal, policy mean 3e-5, every lane cell-exact: rollout_replay.py), and

  * the reset draws (ensemble parameters, x0, r) of the eight slices are pairwise different (no two ranks simulate the same
    plants), and
  * each slice is BIT-EQUAL -- observations, actions, noise, rewards, done flags, final env state -- to lanes
    [g * 16384, (g + 1) * 16384) of ONE 131 072-lane env with the same seed: splitting the lanes over ranks changes nothing.

What stays unmeasured is the 8-GPU hardware itself (RCCL over xGMI): the gradient all-reduce is covered by
tests/test_dist_gloo.py (world 2, gloo) and tests/test_gpu_dp_single_rank.py."""
import numpy as np
import pytest
import torch

from rollout_replay import DEV, make_agent, replay_through_oracle

pytestmark = pytest.mark.gpu
LANES, RANKS, SEED = 16384, 8, 0
ALGO = "ResidualIntegratorModularPPO"


def _collect(env, ag, epoch0):
    """One fused-rollout episode on `env` with the agent's exploration epoch forced to epoch0 + 1 (every rank of a job is at the
    same epoch: they run the same number of rollouts)."""
    from pime_amd.elegantrl.run import make_buffer
    assert ag._fused_rollout_ok(env)
    ag._rollout_epoch = epoch0
    buf = make_buffer(ag, env, env.num_envs * env.max_step)
    assert ag.explore_env(env, buf, env.num_envs * env.max_step, 1.0, 0.99) == env.num_envs * env.max_step
    torch.cuda.synchronize()
    return buf


def test_config4_eight_rank_slices_equal_one_131072_lane_env():
    import oracle
    from pime_amd import gym_control
    table = oracle.ph_table()
    big = gym_control.make_vec(gym_control.PH_V35, LANES * RANKS, device=DEV, state_mode="mixed", seed=SEED)
    ag = make_agent(ALGO, big, 128)
    big_buf = _collect(big, ag, 0)
    T = big.max_step
    fields = ("x", "I", "r", "A", "B", "C", "qww_V", "qc_V")
    big_fields = {f: big.get_field(f) for f in fields}
    first_draws = []
    for g in range(RANKS):
        lo, hi = g * LANES, (g + 1) * LANES
        env = gym_control.make_vec(gym_control.PH_V35, LANES, device=DEV, state_mode="mixed", seed=SEED, env_offset=lo)
        buf = _collect(env, ag, 0)
        # (1) the slice against the oracle with the same lane offset
        ref = oracle.OraclePH(LANES, table, seed=SEED, env_offset=lo)
        first_draws.append(np.stack([buf.state[0, :, 0].cpu().numpy(), buf.state[0, :, 1].cpu().numpy()]))
        replay_through_oracle(ag, ALGO, env, buf, ref, 1, lo, True)
        np.testing.assert_array_equal(env.get_field("qww_V"), ref.get("qww_V"))
        # (2) the slice against lanes [lo, hi) of the one big env: bit for bit
        for name in ("state", "action", "noise", "reward", "done"):
            a, b = getattr(buf, name), getattr(big_buf, name)
            n = T + 1 if name == "state" else T
            assert torch.equal(a[:n], b[:n, lo:hi]), f"rank {g}: {name} differs from the 131072-lane env's lanes [{lo}, {hi})"
        for f in fields:
            np.testing.assert_array_equal(env.get_field(f), big_fields[f][lo:hi], err_msg=f"rank {g}: field {f}")
        env.close()
    # (3) no two ranks drew the same episode: first observations (y0, r) differ lane by lane between any two slices
    for i in range(RANKS):
        for j in range(i + 1, RANKS):
            same = (first_draws[i] == first_draws[j]).all(axis=0).mean()
            assert same < 1e-3, f"ranks {i} and {j} share {same:.4f} of their reset draws"
    big.close()
