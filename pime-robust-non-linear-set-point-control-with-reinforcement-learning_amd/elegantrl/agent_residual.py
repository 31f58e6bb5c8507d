"""Residual agents: the policy output is a correction on top of a fixed prior P/PI controller
(interface of /root/reference/elegantrl/agent_residual.py: Residual :15-24, AgentResidualPPO :31-73,
AgentResidualIntegratorModularPPO :76-95)."""
import numpy as np
import torch
import torch.nn as nn

from .agent import AgentPPO, AgentTD3
from .net import CriticAdv
from .net_residual import ActorResidualIntegratorModularPPO, ActorResidualPPO


class Residual:
    def init_residual(self, residual_kwarg):
        """priorK = -K as a frozen parameter of the actor (checkpointed with it), zero the output layer, and keep a
        float64 host copy for the rollout-side composition (agent_residual.py:16-21)."""
        K = np.asarray(residual_kwarg["init_K"], dtype=np.float64)
        self.act.priorK = nn.Parameter(-torch.as_tensor(K, dtype=torch.float32, device=self.device),
                                       requires_grad=False)
        self.init_actor_zero()
        self.priorK = -K

    def fix_K(self):
        self.act.priorK.requires_grad = False


class AgentResidualPPO(AgentPPO, Residual):
    def _build_nets(self, net_dim, state_dim, action_dim):
        self.cri = CriticAdv(state_dim, net_dim, self.if_use_dn).to(self.device)
        self.act = ActorResidualPPO(net_dim, state_dim, action_dim, self.if_use_dn).to(self.device)

    def _env_action(self, state, action):
        # float32 tanh + float32 state @ float64 priorK, summed in float64 (agent_residual.py:61)
        return np.tanh(action) + state @ self.priorK

    def _vec_env_step(self, env, a_pre, obs, out_obs, out_reward, out_done):
        # the composition above is fused into the env kernel's prologue (binary16 rows: the *_h form of the same kernel)
        step = env.step_residual_h if obs.dtype == torch.float16 else env.step_residual
        return step(a_pre, obs, self.priorK.reshape(-1), auto_reset=True, out_obs=out_obs, out_reward=out_reward,
                    out_done=out_done)

    def _rollout_priorK(self):
        return self.priorK.reshape(-1)

    def frozen_transfer(self):
        self.act.frozen_transfer()
        self.cri.frozen_transfer()


class AgentResidualIntegratorModularPPO(AgentResidualPPO):
    def init(self, net_dim, state_dim, action_dim, integrator_dim, if_per=False):
        self._integrator_dim = integrator_dim
        super().init(net_dim, state_dim, action_dim, if_per)

    def _build_nets(self, net_dim, state_dim, action_dim):
        self.cri = CriticAdv(state_dim, net_dim, self.if_use_dn).to(self.device)
        self.act = ActorResidualIntegratorModularPPO(net_dim, state_dim, action_dim, self._integrator_dim,
                                                     self.if_use_dn).to(self.device)

    def frozen_integrator(self):
        self.act.frozen_integrator()
        self.cri.frozen_transfer()


class AgentResidualTD3(AgentTD3):
    """Residual TD3: BASELINE.json names it, the reference does not have it (`train.py --algo TD3` crashes in
    `init_actor_zero`).  Composed here from the reference's PIECES -- Actor (net.py:96-110), CriticTwin (:305-332), AgentTD3
    (agent.py:276-341), the zero-initialised output layer (agent.py:569-574) and the prior gain priorK = -K
    (agent_residual.py:15-24) -- the way AgentResidualPPO composes its own (agent_residual.py:61):

        env action = clamp(tanh(pi_theta(s)) + explore noise, -1, 1) + s @ priorK        (evaluation: tanh(pi_theta(s)) + s @ priorK)

    The replay buffer and the critics see the RESIDUAL action (before the prior term), as the PPO buffer does.
    No reference oracle exists for the composition; the parts are pinned (tests/test_td3_golden_cpu.py, nets.npz)."""

    def init_residual(self, residual_kwarg):
        K = np.asarray(residual_kwarg["init_K"], dtype=np.float64)
        self.priorK = -K
        self.act.priorK = nn.Parameter(-torch.as_tensor(K, dtype=torch.float32, device=self.device), requires_grad=False)
        self.act_target.priorK = nn.Parameter(self.act.priorK.detach().clone(), requires_grad=False)
        self.init_actor_zero()

    def fix_K(self):
        self.act.priorK.requires_grad = False

    def init_actor_zero(self):
        """Zero the policy's output layer (and its target's): the initial policy is the prior controller alone."""
        with torch.no_grad():
            for net in (self.act, self.act_target):
                net.net[-1].weight.fill_(0.)
                net.net[-1].bias.fill_(0.)
        self._make_optimizers()

    def _prior_term(self, states):
        return states @ self.act.priorK

    def _rollout_priorK(self):
        return self.priorK.reshape(-1)

    def _env_action(self, state, action):
        return action + state @ self.priorK

    def eval_policy(self, states):
        """Deterministic ENV action for evaluation (run.py:600-619 calls the policy on the observation): residual + prior.
        `self.act(states)` alone is the residual action the critics are trained on."""
        return self.act(states) + states @ self.act.priorK
