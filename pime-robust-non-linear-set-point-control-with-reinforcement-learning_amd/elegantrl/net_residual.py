"""Residual actors: a_env = tanh(pi_theta(s)) + s @ priorK, priorK = -K of the prior P/PI controller
(interface of /root/reference/elegantrl/net_residual.py: ActorResidualPPO :6-66,
ActorResidualIntegratorModularPPO :138-205).  The Single/Linear/LinearOut variants are not in the reference's
algorithm map (utils/utils.py:5-11) and the Linear ones cannot be constructed there (net_residual.py:233,298)."""
import torch
import torch.nn as nn

from .net import LOG_SQRT_2PI, GaussianHead, _freeze_all_but_last, layer_norm, mlp


class _ResidualActor(nn.Module, GaussianHead):
    def _finish(self, state_dim, action_dim):
        self.state_dim, self.action_dim = state_dim, action_dim
        self.a_std_log = nn.Parameter(torch.zeros((1, action_dim)) - 0.5, requires_grad=True)   # sigma = e^-0.5
        self.sqrt_2pi_log = LOG_SQRT_2PI
        # overwritten by Residual.init_residual with -K; random until then, as in the reference (:42)
        self.priorK = nn.Parameter(torch.randn(state_dim, action_dim) * 0.01, requires_grad=False)
        layer_norm(self.net[-1], std=0.1)

    def forward(self, state):
        """Deterministic env action (evaluation, run.py:600-619)."""
        return self.mean(state).tanh() + state @ self.priorK


class ActorResidualPPO(_ResidualActor):
    packed_kind = "plain_actor"

    def __init__(self, mid_dim, state_dim, action_dim, if_use_dn=False):
        super().__init__()
        if if_use_dn or not isinstance(state_dim, int):
            raise NotImplementedError("DenseNet / pixel branches are outside the control-env path")
        self.net = mlp([state_dim, mid_dim, mid_dim, mid_dim, action_dim], nn.Tanh)
        self._finish(state_dim, action_dim)

    def mean(self, state):
        return self.net(state)

    def frozen_transfer(self):
        _freeze_all_but_last(self.net)


class ActorResidualIntegratorModularPPO(_ResidualActor):
    """Two towers -- the plant observation [y, r] / [h1, h2, r] and the integrated error -- each
    in -> md Tanh -> md/2 Tanh, concatenated into md -> md Tanh -> A."""
    packed_kind = "modular_actor"

    def __init__(self, mid_dim, state_dim, action_dim, integrator_dim, if_use_dn=False):
        super().__init__()
        if if_use_dn:
            raise NotImplementedError("DenseNet branch is outside the control-env path")
        self.integrator_dim = integrator_dim
        self.other_dim = state_dim - integrator_dim
        half = mid_dim // 2
        self.other_net = mlp([self.other_dim, mid_dim, half], nn.Tanh, out_act=nn.Tanh)
        self.integrator_net = mlp([integrator_dim, mid_dim, half], nn.Tanh, out_act=nn.Tanh)
        self.net = mlp([2 * half, mid_dim, action_dim], nn.Tanh)
        self._finish(state_dim, action_dim)

    def mean(self, state):
        plant = self.other_net(state[:, :self.other_dim])
        integ = self.integrator_net(state[:, self.other_dim:])
        return self.net(torch.cat([plant, integ], dim=-1))

    def frozen_integrator(self):
        for p in self.integrator_net.parameters():
            p.requires_grad = False

    def frozen_transfer(self):
        for tower in (self.integrator_net, self.other_net):
            for p in tower.parameters():
                p.requires_grad = False
        _freeze_all_but_last(self.net)
