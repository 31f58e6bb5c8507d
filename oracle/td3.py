"""CPU restatement (numpy, float64) of one TD3 optimizer step -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the reference line by line (paths under /root/reference):
  * elegantrl/net.py:96-110      Actor: D -> md ReLU -> md ReLU -> md ReLU -> 1; forward = tanh; get_action adds clamp(N(0,1) * std, +-0.5)
                                 and clamps to [-1, 1]
  * elegantrl/net.py:305-332     CriticTwin: cat(state, action) -> md ReLU -> md ReLU, heads net_q1 / net_q2; forward = q1 only
  * elegantrl/agent.py:361-370   get_obj_critic_raw: q_label = reward + mask * min(cri_target.get_q1_q2(next_s, act_target.get_action(next_s)));
                                 obj_critic = SmoothL1(q1, q_label) + SmoothL1(q2, q_label)   (torch default beta = 1, mean)
  * elegantrl/agent.py:314-331   update_net loop body: critic backward + Adam, delayed soft update of cri_target, obj_actor =
                                 -cri_target(state, act(state)).mean() (through the TARGET critic's first head), actor backward + Adam,
                                 delayed soft update of act_target
  * elegantrl/agent.py:116-124   soft_update: tar = cur * tau + tar * (1 - tau)
  * torch.optim.Adam defaults    (agent.py:291,295: lr only): betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad
Pinned by tests/test_oracle_golden.py against tests/golden/td3_update.npz (weights after 6 reference steps) and
tests/golden/td3_update_multi.npz (the reference's own first-step .grad tensors, weights after the first and the last step)."""
import numpy as np

ACTOR_KEYS = ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias", "net.6.weight", "net.6.bias"]
CRITIC_KEYS = ["net_sa.0.weight", "net_sa.0.bias", "net_sa.2.weight", "net_sa.2.bias", "net_q1.weight", "net_q1.bias",
               "net_q2.weight", "net_q2.bias"]


def f64(sd, keys):
    return {k: np.asarray(sd[k], dtype=np.float64).copy() for k in keys}


def actor_hidden(p, s):
    """Hidden activations and the pre-tanh output of Actor.net (net.py:99-102)."""
    h1 = np.maximum(s @ p["net.0.weight"].T + p["net.0.bias"], 0.0)
    h2 = np.maximum(h1 @ p["net.2.weight"].T + p["net.2.bias"], 0.0)
    h3 = np.maximum(h2 @ p["net.4.weight"].T + p["net.4.bias"], 0.0)
    pre = h3 @ p["net.6.weight"].T + p["net.6.bias"]
    return h1, h2, h3, pre


def critic_hidden(p, s, a):
    x = np.concatenate([s, a], axis=1)
    c1 = np.maximum(x @ p["net_sa.0.weight"].T + p["net_sa.0.bias"], 0.0)
    c2 = np.maximum(c1 @ p["net_sa.2.weight"].T + p["net_sa.2.bias"], 0.0)
    q1 = c2 @ p["net_q1.weight"].T + p["net_q1.bias"]
    q2 = c2 @ p["net_q2.weight"].T + p["net_q2.bias"]
    return x, c1, c2, q1, q2


def smooth_l1(d):
    """Elementwise SmoothL1 (beta 1) and its derivative."""
    a = np.abs(d)
    return np.where(a < 1.0, 0.5 * d * d, a - 0.5), np.where(a < 1.0, d, np.sign(d))


def critic_objective(cri, cri_t, act_t, s, a, r, m, s2, eps, policy_noise=0.2, noise_clip=0.5):
    """(obj_critic, gradients of the online critic): agent.py:361-370 + backward()."""
    B = len(s)
    _, _, _, pre = actor_hidden(act_t, s2)
    noise = np.clip(eps.reshape(B, 1) * policy_noise, -noise_clip, noise_clip)          # net.py:109
    next_a = np.clip(np.tanh(pre) + noise, -1.0, 1.0)
    _, _, _, tq1, tq2 = critic_hidden(cri_t, s2, next_a)
    label = r.reshape(B, 1) + m.reshape(B, 1) * np.minimum(tq1, tq2)
    x, c1, c2, q1, q2 = critic_hidden(cri, s, a.reshape(B, 1))
    l1, g1 = smooth_l1(q1 - label)
    l2, g2 = smooth_l1(q2 - label)
    obj = l1.mean() + l2.mean()
    g1, g2 = g1 / B, g2 / B
    g = {"net_q1.weight": g1.T @ c2, "net_q1.bias": g1.sum(0), "net_q2.weight": g2.T @ c2, "net_q2.bias": g2.sum(0)}
    dz2 = (g1 @ cri["net_q1.weight"] + g2 @ cri["net_q2.weight"]) * (c2 > 0)
    g["net_sa.2.weight"], g["net_sa.2.bias"] = dz2.T @ c1, dz2.sum(0)
    dz1 = (dz2 @ cri["net_sa.2.weight"]) * (c1 > 0)
    g["net_sa.0.weight"], g["net_sa.0.bias"] = dz1.T @ x, dz1.sum(0)
    return obj, g


def actor_objective(act, cri_t, s):
    """(obj_actor, gradients of the actor): agent.py:323-326, -mean(cri_target(state, act(state))) differentiated through the target
    critic's trunk and first head into the actor."""
    B = len(s)
    h1, h2, h3, pre = actor_hidden(act, s)
    action = np.tanh(pre)
    _, c1, c2, q1, _ = critic_hidden(cri_t, s, action)
    obj = -q1.mean()
    gq = np.full((B, 1), -1.0 / B)
    dz2 = (gq @ cri_t["net_q1.weight"]) * (c2 > 0)
    dz1 = (dz2 @ cri_t["net_sa.2.weight"]) * (c1 > 0)
    da = dz1 @ cri_t["net_sa.0.weight"][:, -1:]                                           # the action column
    dpre = da * (1.0 - action * action)
    g = {"net.6.weight": dpre.T @ h3, "net.6.bias": dpre.sum(0)}
    d3 = (dpre @ act["net.6.weight"]) * (h3 > 0)
    g["net.4.weight"], g["net.4.bias"] = d3.T @ h2, d3.sum(0)
    d2 = (d3 @ act["net.4.weight"]) * (h2 > 0)
    g["net.2.weight"], g["net.2.bias"] = d2.T @ h1, d2.sum(0)
    d1 = (d2 @ act["net.2.weight"]) * (h1 > 0)
    g["net.0.weight"], g["net.0.bias"] = d1.T @ s, d1.sum(0)
    return obj, g


class Adam:
    """torch.optim.Adam with its defaults, per tensor."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, betas[0], betas[1], eps, 0
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        bc1, bc2 = 1.0 - self.b1 ** self.t, 1.0 - self.b2 ** self.t
        for k, g in grads.items():
            g = g.reshape(params[k].shape)
            self.m[k] += (g - self.m[k]) * (1.0 - self.b1)
            self.v[k] = self.v[k] * self.b2 + g * g * (1.0 - self.b2)
            params[k] -= (self.lr / bc1) * (self.m[k] / (np.sqrt(self.v[k]) / np.sqrt(bc2) + self.eps))


def soft_update(tar, cur, tau):
    for k in tar:
        tar[k] = cur[k] * tau + tar[k] * (1.0 - tau)


class Td3:
    """The four nets and two optimizers of AgentTD3 (agent.py:283-296), float64."""

    def __init__(self, act, act_t, cri, cri_t, lr=1e-4, tau=2 ** -8, policy_noise=0.2, update_freq=2):
        self.act, self.act_t = f64(act, ACTOR_KEYS), f64(act_t, ACTOR_KEYS)
        self.cri, self.cri_t = f64(cri, CRITIC_KEYS), f64(cri_t, CRITIC_KEYS)
        self.opt_a, self.opt_c = Adam(self.act, lr), Adam(self.cri, lr)
        self.tau, self.policy_noise, self.update_freq = tau, policy_noise, update_freq

    def step(self, i, state, other, idx, nxt, eps):
        """Iteration i of update_net's loop on the sampled rows idx (successors nxt) with the smoothing-noise draws eps.
        Returns (obj_actor, obj_critic, critic gradients, actor gradients)."""
        s, s2 = state[idx].astype(np.float64), state[nxt].astype(np.float64)
        o = other[idx].astype(np.float64)
        obj_c, gc = critic_objective(self.cri, self.cri_t, self.act_t, s, o[:, 2], o[:, 0], o[:, 1], s2,
                                     np.asarray(eps, dtype=np.float64), self.policy_noise)
        self.opt_c.step(self.cri, gc)
        soft = i % self.update_freq == 0
        if soft:
            soft_update(self.cri_t, self.cri, self.tau)
        obj_a, ga = actor_objective(self.act, self.cri_t, s)
        self.opt_a.step(self.act, ga)
        if soft:
            soft_update(self.act_t, self.act, self.tau)
        return obj_a, obj_c, gc, ga


def smoothing_noise(seed, epoch, row, B):
    """float32[B]: the draws the fused step makes when no noise table is given -- Philox4x32-10 keyed by `seed`, counter (batch
    position, epoch, table row, stream 3), Box-Muller cosine branch (csrc/td3_fused.hip: td3_noise evaluates log / sqrt / cos with the
    float32 hardware transcendentals; this float64 evaluation of the same uniforms agrees to ~1e-6 relative, far inside the gradient
    tolerance of the test that uses it)."""
    from .binding import philox_uniform_pair
    out = np.empty(B, dtype=np.float32)
    for p in range(B):
        ua, ub = philox_uniform_pair(seed, p, epoch, row, 3)
        out[p] = np.float32(np.sqrt(-2.0 * np.log(1.0 - ua)) * np.cos(6.283185307179586476925286766559 * ub))
    return out
