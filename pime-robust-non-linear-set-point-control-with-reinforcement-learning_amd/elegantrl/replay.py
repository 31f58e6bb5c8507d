"""Experience storage.

`ReplayBuffer` keeps the reference's flat ring interface (/root/reference/elegantrl/replay.py:238-433) for the
one-instance env path and the off-policy agents: rows are transitions in time order, `buf_other` = (reward*scale,
mask, action[, noise]) and `sample_batch` relies on row i+1 being the successor of row i.  Unlike the reference it
stays on the agent's device also for on-policy use (the reference forces host NumPy there and re-uploads the whole
buffer every update, replay.py:265-266,353-367).

`VecReplayBuffer` is the off-policy ring for N lock-stepped env lanes: time-major [L, N, .] in HBM; the successor of
(slot t, lane n) is (slot t+1, lane n), i.e. flat row i + N where the reference uses i + 1 (replay.py:344-351).

`TrajectoryBuffer` is the vectorised-rollout store: time-major [T, N, .] tensors resident in HBM that the env
step kernel and the policy write into directly; lane n of slot t is the transition env n made at its step t.
"""
import numpy as np
import torch


class ReplayBuffer:
    def __init__(self, max_len, state_dim, action_dim, if_on_policy, if_per=False, if_gpu=True, device=None):
        if if_per:
            raise NotImplementedError("prioritised replay is not on the residual-control path (run.py:37 if_per=False)")
        self.device = torch.device(device) if device is not None else \
            torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.max_len = int(max_len)
        self.now_len = 0
        self.next_idx = 0
        self.if_full = False
        self.action_dim = action_dim
        self.if_on_policy = if_on_policy
        self.if_gpu = True
        self.if_per = False
        other_dim = 2 + action_dim * (2 if if_on_policy else 1)
        self.buf_other = torch.empty((self.max_len, other_dim), dtype=torch.float32, device=self.device)
        self.buf_state = torch.empty((self.max_len, state_dim), dtype=torch.float32, device=self.device)

    def append_buffer(self, state, other):
        self.buf_state[self.next_idx] = torch.as_tensor(state, dtype=torch.float32, device=self.device)
        self.buf_other[self.next_idx] = torch.as_tensor(np.asarray(other, dtype=np.float32), device=self.device)
        self.next_idx += 1
        if self.next_idx >= self.max_len:
            self.if_full = True
            self.next_idx = 0

    def extend_buffer(self, state, other):
        state = torch.as_tensor(state, dtype=torch.float32, device=self.device)
        other = torch.as_tensor(other, dtype=torch.float32, device=self.device)
        size = len(other)
        end = self.next_idx + size
        if end > self.max_len:  # wrap: head of the batch fills the tail of the ring, the rest restarts at row 0
            head = self.max_len - self.next_idx
            self.buf_state[self.next_idx:] = state[:head]
            self.buf_other[self.next_idx:] = other[:head]
            self.if_full = True
            end -= self.max_len
            self.buf_state[:end] = state[-end:] if end else state[:0]
            self.buf_other[:end] = other[-end:] if end else other[:0]
        else:
            self.buf_state[self.next_idx:end] = state
            self.buf_other[self.next_idx:end] = other
        self.next_idx = end

    def sample_batch(self, batch_size, indices=None):
        """(reward, mask, action, state, next_state): next_state is row i+1 (time adjacency)."""
        if indices is None:
            indices = torch.randint(self.now_len - 1, size=(batch_size,), device=self.device)
        r_m_a = self.buf_other[indices]
        return r_m_a[:, 0:1], r_m_a[:, 1:2], r_m_a[:, 2:], self.buf_state[indices], self.buf_state[indices + 1]

    def sample_all(self):
        """(reward, mask, action, noise, state) of every stored on-policy transition, in storage order."""
        other = self.buf_other[:self.now_len]
        a = self.action_dim
        return other[:, 0], other[:, 1], other[:, 2:2 + a], other[:, 2 + a:], self.buf_state[:self.now_len]

    def update_now_len_before_sample(self):
        self.now_len = self.max_len if self.if_full else self.next_idx

    def empty_buffer_before_explore(self):
        self.next_idx = 0
        self.now_len = 0
        self.if_full = False


class TrajectoryBuffer:
    """Time-major on-policy store for N lock-stepped env lanes: state [T,N,D], reward/mask [T,N], action/noise [T,N,A]."""

    def __init__(self, horizon, num_envs, state_dim, action_dim, device, dtype=torch.float32):
        """dtype: storage type of the observation and reward rows -- float32, or float16 for an env in state_mode "mixed16"
        (BASELINE.json config 5: the env kernels write binary16 rows; `sample_all` widens them once per update, the PPO
        kernels compute in float32).  Actions, noise and masks stay float32."""
        self.device = torch.device(device)
        self.horizon, self.num_envs, self.state_dim, self.action_dim = horizon, num_envs, state_dim, action_dim
        f = dict(dtype=torch.float32, device=self.device)
        self.dtype = dtype
        # one extra state slot: the step kernel writes the successor observation of slot t straight into slot t+1
        self.state = torch.zeros((horizon + 1, num_envs, state_dim), dtype=dtype, device=self.device)
        self.reward = torch.zeros((horizon, num_envs), dtype=dtype, device=self.device)
        self.mask = torch.zeros((horizon, num_envs), **f)
        self.action = torch.zeros((horizon, num_envs, action_dim), **f)
        self.noise = torch.zeros((horizon, num_envs, action_dim), **f)
        self.done = torch.zeros((horizon, num_envs), dtype=torch.uint8, device=self.device)
        self.length = 0  # filled time slots
        self.if_on_policy = True

    @property
    def now_len(self):
        return self.length * self.num_envs

    def update_now_len_before_sample(self):
        pass

    def empty_buffer_before_explore(self):
        self.length = 0

    def sample_all(self):
        T = self.length
        flat = lambda x: x[:T].reshape(T * self.num_envs, *x.shape[2:])  # noqa: E731
        wide = (lambda x: x.float()) if self.dtype != torch.float32 else (lambda x: x)   # binary16 rows: widened once per update
        return wide(flat(self.reward)), flat(self.mask), flat(self.action), flat(self.noise), wide(flat(self.state))


class VecReplayBuffer:
    """Off-policy ring buffer for N lock-stepped env lanes, resident on the device.

    Slot t holds what every lane did at its t-th stored step: state [L, N, D], other [L, N, 2 + A] = (reward * scale, mask,
    action).  The flat views `buf_state` [L * N, D] / `buf_other` keep the reference's row semantics with one change: the
    successor of row i is row i + N (same lane, next slot), not i + 1 (replay.py:350: `self.buf_state[indices + 1]` relies on one
    env filling the ring in time order).  A lane that ended an episode at slot t has mask 0 there and the first observation
    of its next episode at slot t + 1, exactly as the reference stores `env.reset()` in the next row.

    `sample_batch` draws (slot, lane) uniformly over the stored slots THAT HAVE a successor: the newest slot is excluded --
    when the ring is full its "successor" would be the oldest slot (the reference samples that row too, replay.py:345 draws
    from now_len - 1 rows regardless of where the write cursor is; a documented deviation, 1 / L of the rows)."""

    def __init__(self, max_len, num_envs, state_dim, action_dim, device):
        self.device = torch.device(device)
        self.num_envs, self.state_dim, self.action_dim = int(num_envs), int(state_dim), int(action_dim)
        self.slots = max(2, -(-int(max_len) // self.num_envs))
        self.max_len = self.slots * self.num_envs
        f = dict(dtype=torch.float32, device=self.device)
        self.state = torch.zeros((self.slots, self.num_envs, self.state_dim), **f)
        self.other = torch.zeros((self.slots, self.num_envs, 2 + self.action_dim), **f)
        self.buf_state = self.state.view(self.max_len, self.state_dim)
        self.buf_other = self.other.view(self.max_len, 2 + self.action_dim)
        self.next_slot = 0
        self.if_full = False
        self.now_len = 0
        self.if_on_policy = False
        self._bounds = torch.zeros(2, dtype=torch.int64, device=self.device)   # see update_now_len_before_sample

    @property
    def stored_slots(self):
        return self.slots if self.if_full else self.next_slot

    def append_step(self, state, reward, mask, action):
        """One lock-step of every lane: state [N, D], reward / mask [N], action [N, A] (device tensors)."""
        t = self.next_slot
        self.state[t].copy_(state)
        o = self.other[t]
        o[:, 0].copy_(reward)
        o[:, 1].copy_(mask)
        o[:, 2:].copy_(action.reshape(self.num_envs, self.action_dim))
        self.next_slot += 1
        if self.next_slot >= self.slots:
            self.next_slot, self.if_full = 0, True

    def advance(self, n):
        """`n` lock-steps were written into slots next_slot .. next_slot + n - 1 (mod slots) by the fused exploration kernel."""
        assert 1 <= n <= self.slots
        if self.next_slot + n >= self.slots:
            self.if_full = True
        self.next_slot = (self.next_slot + n) % self.slots

    def update_now_len_before_sample(self):
        """Also publishes the sampler's bounds to the device (rows that have a successor, the oldest slot): `sample_indices`
        reads them there, so an update captured into a HIP graph keeps sampling the right rows as the ring fills and its
        cursor moves -- the graph survives from one `update_net` call to the next."""
        self.now_len = self.stored_slots * self.num_envs
        n_slots = self.stored_slots
        host = torch.tensor([max(n_slots - 1, 0) * self.num_envs, self.next_slot if self.if_full else 0], dtype=torch.int64)
        self._bounds.copy_(host)

    def sample_indices(self, batch_size, out=None):
        """Flat row indices [batch] of transitions with a valid successor, and their successors' indices.  Uniform over the
        (stored_slots - 1) * N rows whose successor slot is stored: a 62-bit draw reduced modulo the row count that lives on
        the device (bias < 2^-40; torch.randint(high) would bake `high` into a captured graph)."""
        assert self.stored_slots >= 2, "need two stored steps before sampling"
        N = self.num_envs
        u = torch.randint(2 ** 62, size=(batch_size,), device=self.device) if out is None else \
            torch.randint(2 ** 62, size=(batch_size,), device=self.device, out=out)
        u = u % self._bounds[0]
        lane = u % N
        slot = (u // N + self._bounds[1]) % self.slots   # slots in age order start at the oldest (the write cursor once full)
        return slot * N + lane, ((slot + 1) % self.slots) * N + lane

    def cut_last_step(self):
        """The newest stored lock-step gets mask 0: called when its successor slot will NOT hold the lanes' next observation
        (the env was reset by someone else in between), so no target bootstraps across the cut."""
        if self.stored_slots:
            self.other[(self.next_slot - 1) % self.slots, :, 1] = 0.0

    def sample_batch(self, batch_size):
        """(reward, mask, action, state, next_state), shapes as ReplayBuffer.sample_batch."""
        idx, nxt = self.sample_indices(batch_size)
        r_m_a = self.buf_other[idx]
        return r_m_a[:, 0:1], r_m_a[:, 1:2], r_m_a[:, 2:], self.buf_state[idx], self.buf_state[nxt]

    def empty_buffer_before_explore(self):
        self.next_slot, self.if_full, self.now_len = 0, False, 0
