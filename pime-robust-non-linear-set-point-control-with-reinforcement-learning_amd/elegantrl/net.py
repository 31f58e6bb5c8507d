"""Base actor / critic MLPs (interface of /root/reference/elegantrl/net.py for the classes the residual path
reaches: Actor :96-110, ActorPPO :113-172, CriticAdv :256-302, CriticTwin :305-332, layer_norm :617-619).

state_dict key layout is the reference's (`net.0.weight`, `net.2.bias`, ..., `a_std_log`), so checkpoints written
by either side load in the other (agent.py:86-114).  Only the vector-state (int state_dim) variants exist here:
the pixel/conv and DenseNet branches are never reached from train.py with the control envs.
"""
import math

import torch
import torch.nn as nn

LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


def layer_norm(layer, std=1.0, bias_const=1e-6):
    """Orthogonal weight of gain `std`, constant bias (net.py:617-619)."""
    nn.init.orthogonal_(layer.weight, std)
    nn.init.constant_(layer.bias, bias_const)


def mlp(sizes, hidden_act, out_act=None):
    """Linear layers at even indices, activations at odd ones -- the indexing the reference checkpoints use."""
    mods = []
    for i in range(len(sizes) - 1):
        mods.append(nn.Linear(sizes[i], sizes[i + 1]))
        last = i == len(sizes) - 2
        if not last:
            mods.append(hidden_act())
        elif out_act is not None:
            mods.append(out_act())
    return nn.Sequential(*mods)


def _freeze_all_but_last(seq):
    for p in seq.parameters():
        p.requires_grad = False
    for p in seq[-1].parameters():
        p.requires_grad = True


class GaussianHead:
    """Shared PPO exploration / log-likelihood maths on top of a `mean(state)` (net.py:150-164)."""

    def get_action_noise(self, state, noise=None):
        a_avg = self.mean(state)
        if noise is None:
            noise = torch.randn_like(a_avg)
        return a_avg + noise * self.a_std_log.exp(), noise

    def compute_logprob(self, state, action):
        a_avg = self.mean(state)
        delta = ((a_avg - action) / self.a_std_log.exp()).pow(2) * 0.5
        return -(self.a_std_log + self.sqrt_2pi_log + delta).sum(1)

    def old_logprob(self, noise):
        """log-prob of the action that was sampled with `noise` (agent.py:621)."""
        return -(noise.pow(2) * 0.5 + self.a_std_log + self.sqrt_2pi_log).sum(1)


class Actor(nn.Module):
    """Deterministic policy of TD3/DDPG: D -> md ReLU -> md ReLU -> md ReLU -> A, tanh-squashed."""
    packed_kind = "critic"   # image kind of its pre-tanh mean for the fused kernels: CriticAdv's shape and activations (A = 1)

    def __init__(self, mid_dim, state_dim, action_dim):
        super().__init__()
        self.state_dim, self.action_dim = state_dim, action_dim
        self.net = mlp([state_dim, mid_dim, mid_dim, mid_dim, action_dim], nn.ReLU)

    def forward(self, state):
        return self.net(state).tanh()

    def get_action(self, state, action_std):
        action = self.net(state).tanh()
        noise = (torch.randn_like(action) * action_std).clamp(-0.5, 0.5)
        return (action + noise).clamp(-1.0, 1.0)


class ActorPPO(nn.Module, GaussianHead):
    """Stochastic policy: D -> md Tanh -> md Tanh -> md Tanh -> A with a state-independent log-std."""
    packed_kind = "plain_actor"

    def __init__(self, mid_dim, state_dim, action_dim, if_use_dn=False):
        super().__init__()
        if if_use_dn or not isinstance(state_dim, int):
            raise NotImplementedError("DenseNet / pixel branches are outside the control-env path")
        self.state_dim, self.action_dim = state_dim, action_dim
        self.net = mlp([state_dim, mid_dim, mid_dim, mid_dim, action_dim], nn.Tanh)
        self.a_std_log = nn.Parameter(torch.zeros((1, action_dim)) - 0.5, requires_grad=True)
        self.sqrt_2pi_log = LOG_SQRT_2PI
        layer_norm(self.net[-1], std=0.1)

    def mean(self, state):
        return self.net(state)

    def forward(self, state):
        return self.net(state).tanh()

    def frozen_transfer(self):
        _freeze_all_but_last(self.net)


class CriticAdv(nn.Module):
    """State-value net.  The reference builds a Hardswish variant and then overwrites it (net.py:269-277): the
    effective net is D -> md ReLU -> md ReLU -> md ReLU -> 1, output layer orthogonal with gain 0.5."""
    packed_kind = "critic"

    def __init__(self, state_dim, mid_dim, if_use_dn=False):
        super().__init__()
        if if_use_dn or not isinstance(state_dim, int):
            raise NotImplementedError("DenseNet / pixel branches are outside the control-env path")
        self.state_dim = state_dim
        # The reference constructs (and discards) a first stack before this one; building a throw-away stack of
        # the same shapes keeps the torch RNG stream -- hence same-seed initial weights -- identical to it.
        mlp([state_dim, mid_dim, mid_dim, mid_dim, 1], nn.ReLU)
        self.net = mlp([state_dim, mid_dim, mid_dim, mid_dim, 1], nn.ReLU)
        layer_norm(self.net[-1], std=0.5)

    def forward(self, state):
        return self.net(state)

    def frozen_transfer(self):
        _freeze_all_but_last(self.net)


class CriticTwin(nn.Module):
    """Shared trunk (D+A -> md ReLU -> md ReLU) with two Q heads (TD3)."""

    def __init__(self, mid_dim, state_dim, action_dim, if_use_dn=False):
        super().__init__()
        if if_use_dn:
            raise NotImplementedError("DenseNet branch is outside the control-env path")
        self.net_sa = mlp([state_dim + action_dim, mid_dim, mid_dim], nn.ReLU, out_act=nn.ReLU)
        self.net_q1 = nn.Linear(mid_dim, 1)
        self.net_q2 = nn.Linear(mid_dim, 1)
        layer_norm(self.net_q1, std=0.1)
        layer_norm(self.net_q2, std=0.1)

    def forward(self, state, action):
        return self.net_q1(self.net_sa(torch.cat((state, action), dim=1)))

    def get_q1_q2(self, state, action):
        trunk = self.net_sa(torch.cat((state, action), dim=1))
        return self.net_q1(trunk), self.net_q2(trunk)
