"""Tensor-level wrappers over the trajectory / MLP kernels of libpime_hip.so.

All operands are CUDA (ROCm) tensors; launches go to torch's current stream.  Nothing here has a CPU path:
a CPU tensor raises.
"""
import ctypes as C
import os

import torch

from . import native

_KINDS = {"critic": native.MLP_CRITIC, "plain_actor": native.MLP_PLAIN_ACTOR, "modular_actor": native.MLP_MODULAR_ACTOR}
_PARAM_ORDER = {
    "critic": ["net.0", "net.2", "net.4", "net.6"],
    "plain_actor": ["net.0", "net.2", "net.4", "net.6"],
    "modular_actor": ["other_net.0", "other_net.2", "integrator_net.0", "integrator_net.2", "net.0", "net.2"],
}


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise native.PimeError("pime_amd.ops kernels need GPU tensors (no CPU fallback)")


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def gae_scan(reward, mask, value, lam, use_gae=True, out_r_sum=None, out_adv=None):
    """ElegantRL's reward-sum / advantage recursion over a time-major [T, N] float32 buffer
    (replaces elegantrl/agent.py:666-708).  Returns un-normalised (r_sum, adv), both [T, N]."""
    _need_cuda(reward, mask, value)
    T, N = reward.shape
    reward, mask, value = (x.contiguous() for x in (reward, mask, value))
    assert reward.dtype == mask.dtype == value.dtype == torch.float32 and mask.shape == value.shape == (T, N)
    r_sum = torch.empty_like(reward) if out_r_sum is None else out_r_sum
    adv = torch.empty_like(reward) if out_adv is None else out_adv
    with torch.cuda.device(reward.device):
        native.check(native.lib().pime_gae_scan(native.ptr(reward), native.ptr(mask), native.ptr(value), T, N,
                                                C.c_float(lam), int(bool(use_gae)), native.ptr(r_sum), native.ptr(adv),
                                                _stream(reward)), "pime_gae_scan")
    return r_sum, adv


class PackedMLP:
    """A small MLP re-laid for the fused f32-MFMA forward kernel (csrc/mlp_mfma.hip).

    `repack()` must be called after the source weights change (it is one tiny launch); `__call__` runs the whole
    net per 32-row tile with activations in registers and returns the scalar head, shape [M]."""

    def __init__(self, kind, state_dim, integrator_dim, mid_dim, device):
        self.kind, self.D, self.Di, self.md = kind, int(state_dim), int(integrator_dim), int(mid_dim)
        self.device = torch.device(device)
        n = native.lib().pime_mlp_packed_floats(_KINDS[kind], self.D, self.Di, self.md)
        if n <= 0:
            raise native.PimeError(f"fused MLP forward unsupported for {kind} D={state_dim} width={mid_dim}: "
                                   f"{native.last_error()}")
        self.packed = torch.empty(n, dtype=torch.float32, device=self.device)
        self._src = None

    @staticmethod
    def supported(kind, state_dim, integrator_dim, mid_dim):
        return native.lib().pime_mlp_packed_floats(_KINDS[kind], int(state_dim), int(integrator_dim), int(mid_dim)) > 0

    @classmethod
    def from_state_dict(cls, kind, sd, state_dim, integrator_dim=0, prefix=""):
        names = _PARAM_ORDER[kind]
        md = sd[f"{prefix}{names[0]}.weight"].shape[0]
        dev = sd[f"{prefix}{names[0]}.weight"].device
        self = cls(kind, state_dim, integrator_dim, md, dev)
        self._src = [sd[f"{prefix}{n}.{p}"] for n in names for p in ("weight", "bias")]
        self.repack()
        return self

    @classmethod
    def from_module(cls, module):
        """module: CriticAdv / ActorResidualPPO / ActorPPO / ActorResidualIntegratorModularPPO of this package."""
        kind = module.packed_kind
        sd = dict(module.named_parameters())
        return cls.from_state_dict(kind, sd, module.state_dim, getattr(module, "integrator_dim", 0))

    def repack(self):
        srcs = []
        for t in self._src:
            t = t.detach()
            _need_cuda(t)
            assert t.dtype == torch.float32
            srcs.append(t.contiguous())
        arr = (C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
        with torch.cuda.device(self.device):
            native.check(native.lib().pime_mlp_pack(_KINDS[self.kind], self.D, self.Di, self.md, arr,
                                                    native.ptr(self.packed), _stream(self.packed)), "pime_mlp_pack")
        return self

    def __call__(self, x, out=None):
        _need_cuda(x)
        x = x.contiguous()
        assert x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == self.D, (x.shape, self.D)
        M = x.shape[0]
        out = torch.empty(M, dtype=torch.float32, device=x.device) if out is None else out
        with torch.cuda.device(x.device):
            native.check(native.lib().pime_mlp_forward(_KINDS[self.kind], native.ptr(x), M, self.D, self.Di, self.md,
                                                       native.ptr(self.packed), native.ptr(out), _stream(x)),
                         "pime_mlp_forward")
        return out


class FusedPPOGrad:
    """Minibatch loss gradients of (actor, critic) by the fused HIP kernels (csrc/ppo_train.hip), written straight
    into the parameters' .grad tensors.  Everything between drawing the minibatch indices and `optimizer.step()`.

    All .grad tensors are views into ONE flat buffer (`self.flat_grad`), so zeroing is one memset and a data-parallel
    all-reduce is one collective on that buffer."""

    def __init__(self, act, cri, max_batch):
        self.act, self.cri = act, cri
        self.device = next(cri.parameters()).device
        _need_cuda(next(cri.parameters()))
        if getattr(act, "action_dim", 1) != 1:
            raise native.PimeError("fused PPO gradients support action_dim == 1")
        self.nets = []
        self.max_batch = int(max_batch)
        L = native.lib()
        # flat gradient buffer over every trainable parameter, in optimizer order (actor first, then critic)
        params = [p for p in list(act.parameters()) + list(cri.parameters()) if p.requires_grad]
        self.params = params
        total = sum(p.numel() for p in params)
        # four more words behind the gradients: under data parallelism the minibatch's target moments (sum r_sum, sum r_sum^2, B)
        # ride in the SAME all-reduce as the gradients (flat_grad_dp), so that the critic scale of agent.py:652 is that of the
        # union minibatch (pime_adam_step_dp)
        self.flat_grad_dp = torch.zeros(total + 4, dtype=torch.float32, device=self.device)
        self.flat_grad = self.flat_grad_dp[:total]
        self.dp_moments = self.flat_grad_dp[total:]
        self.critic_offset = sum(p.numel() for p in act.parameters() if p.requires_grad)   # flat order: actor, then critic
        off = 0
        for p in params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        # The parameters themselves are re-homed into ONE flat tensor as well (each nn.Parameter becomes a view of it),
        # so the optimizer sees a single 67k-element tensor: torch's fused Adam then runs one small launch instead
        # of a 22-tensor multi-tensor apply (27 us -> ~6 us per step on MI355X).
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=self.device)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + n].view_as(p)
                off += n
        self.flat_param.grad = self.flat_grad
        self._dump = torch.zeros(max(p.numel() for p in list(act.parameters()) + list(cri.parameters())),
                                 dtype=torch.float32, device=self.device)  # sink for frozen parameters' gradients
        self.loss_sums = torch.zeros(6, dtype=torch.float32, device=self.device)   # include/pime_hip.h: pime_ppo_minibatch_grad
        self.moments = torch.zeros(2, dtype=torch.float64, device=self.device)
        for module in (act, cri):
            kind = module.packed_kind
            names = _PARAM_ORDER[kind]
            sd = dict(module.named_parameters())
            plist = [sd[f"{n}.{p}"] for n in names for p in ("weight", "bias")]
            D, Di = module.state_dim, getattr(module, "integrator_dim", 0)
            md = plist[0].shape[0]
            k = _KINDS[kind]
            n_fwd = L.pime_ppo_fwd_image_floats(k, D, Di, md)
            n_bwd = L.pime_ppo_bwd_image_floats(k, D, Di, md)
            n_ws = L.pime_ppo_workspace_floats(k, self.max_batch, md)
            if n_fwd <= 0 or n_bwd <= 0 or n_ws <= 0:
                raise native.PimeError(f"fused PPO gradients unsupported for {kind} width {md}: {native.last_error()}")
            net = dict(kind=k, D=D, Di=Di, md=md, plist=plist,
                       n_bwd_f32=L.pime_ppo_bwd_image_f32_floats(k, D, Di, md),   # < n_bwd under PIME_GRAD_BF16X3=1: bf16 planes follow
                       img_fwd=torch.zeros(n_fwd, dtype=torch.float32, device=self.device),   # zeros: the padding the pack
                       img_bwd=torch.zeros(n_bwd, dtype=torch.float32, device=self.device),   # kernels never write is defined
                       ws=torch.empty(n_ws, dtype=torch.float32, device=self.device))
            self.nets.append(net)
        self._structs = None
        self._image_map = None
        self.repack()

    def image_map(self):
        """int32 [2 n]: where every element of the flat parameter tensor sits in its net's forward / transposed image
        (pime_ppo_image_map).  With it the fused optimizer step keeps the images current by itself (`images_follow_step`) and the
        re-pack launch after every step goes away; None if the library cannot derive it (the caller then re-packs as before).
        PIME_NO_IMAGE_MAP=1 turns it off (A/B)."""
        if self._image_map is None:
            if os.environ.get("PIME_NO_IMAGE_MAP"):
                self._image_map = False
                return None
            if self._structs is None:
                self._build_structs()
            actor, critic, _ = self._structs
            m = torch.empty(2 * self.flat_param.numel(), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                rc = native.lib().pime_ppo_image_map(C.byref(actor), C.byref(critic), self.flat_param.data_ptr(),
                                                     self.flat_param.numel(), m.data_ptr(), _stream(self.flat_param))
            self._image_map = m if rc == 0 else False
            self.image_map_error = None if rc == 0 else native.last_error()
            if rc == 0 and not self._image_map_is_exact(m):
                # the map assumes every pack kernel is a pure permutation of the parameters; a pack kernel that ever scales or
                # folds a weight would make a map-scattered image differ from a re-packed one: then re-pack after every step
                self._image_map, self.image_map_error = False, "map-scattered images differ from the re-packed ones"
        return self._image_map if self._image_map is not False else None

    def _image_map_is_exact(self, m):
        """Scatter the CURRENT parameter values through the map into zeroed images and compare them bit for bit with what
        repack() laid down: every mapped position must hold its parameter, every unmapped one the packing's zero padding."""
        self.repack()
        pairs = m.view(-1, 2).to(torch.int64)
        for col, key in ((0, "img_fwd"), (1, "img_bwd")):
            code = pairs[:, col]
            valid = code >= 0
            pos, which = code & ((1 << 28) - 1), (code >> 28) & 3
            for net_bit, net in ((1, self.nets[0]), (0, self.nets[1])):     # bits 28..29: 0 critic, 1 actor; self.nets = [actor, critic]
                sel = valid & (which == net_bit)
                have = net[key] if col == 0 else net[key][:net["n_bwd_f32"]]   # the part that is a permutation of the parameters
                if bool((pos[sel] >= have.numel()).any()):
                    return False
                if all(p.requires_grad for p in net["plist"]):   # every image element is a mapped parameter or zero padding
                    want = torch.zeros_like(have)
                    want[pos[sel]] = self.flat_param.detach()[sel]
                    if not torch.equal(want, have):
                        return False
                elif not torch.equal(have[pos[sel]], self.flat_param.detach()[sel]):   # frozen tensors are in the images, not the map
                    return False
        return True

    @property
    def images_follow_step(self):
        """True: a fused step (`adam=` given to __call__) also updates the packed images; no repack() needed after it."""
        return self.image_map() is not None

    @staticmethod
    def supported(act, cri):
        try:
            for m in (act, cri):
                kind = getattr(m, "packed_kind", None)
                if kind is None or getattr(m, "action_dim", 1) != 1:
                    return False
                sd = dict(m.named_parameters())
                md = sd[_PARAM_ORDER[kind][0] + ".weight"].shape[0]
                if not PackedMLP.supported(kind, m.state_dim, getattr(m, "integrator_dim", 0), md):
                    return False
            return next(cri.parameters()).is_cuda
        except Exception:
            return False

    def _ptr_array(self, tensors):
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def repack(self):
        """Re-lay both nets' weights into the forward / transposed LDS images (after every optimizer step): one launch."""
        if self._structs is None:
            self._build_structs()
        actor, critic, _ = self._structs
        with torch.cuda.device(self.device):
            native.check(native.lib().pime_ppo_repack(C.byref(actor), C.byref(critic), _stream(self.flat_grad)),
                         "pime_ppo_repack")

    def _build_structs(self):
        out = []
        keep = []
        for net, module in zip(self.nets, (self.act, self.cri)):
            grads = [p.grad if (p.requires_grad and p.grad is not None) else self._dump for p in net["plist"]]
            pa, ga = self._ptr_array([p.detach() for p in net["plist"]]), self._ptr_array(grads)
            keep += [pa, ga]
            st = native.PpoNet(kind=net["kind"], D=net["D"], Di=net["Di"], md=net["md"],
                               params=C.cast(pa, C.c_void_p), grads=C.cast(ga, C.c_void_p),
                               img_fwd=net["img_fwd"].data_ptr(), img_bwd=net["img_bwd"].data_ptr(),
                               workspace=net["ws"].data_ptr())
            if module is self.act:
                asl = module.a_std_log
                g = asl.grad if (asl.requires_grad and asl.grad is not None) else self._dump
                st.a_std_log, st.g_a_std_log = asl.data_ptr(), g.data_ptr()
            out.append(st)
        self._structs = (out[0], out[1], keep)

    def make_optimizer(self, lr):
        """Adam over the single flat parameter tensor (same lr for both nets, no weight decay: agent.py:565-566)."""
        return FlatAdam(self.flat_param, self.flat_grad, lr)

    def params_are(self, agent):
        """True while this object still wraps the agent's current nets (they are rebuilt by agent.init)."""
        return agent.act is self.act and agent.cri is self.cri

    def zero_grad(self):
        self.flat_grad.zero_()

    def __call__(self, state, action, logprob, adv, r_sum, indices, ratio_clip, lambda_entropy, critic_scale,
                 overwrite=False, index_row=None, adam=None, defer_critic_scale=False):
        """Accumulates (overwrite=True: writes) d(obj_united)/d(theta) of the minibatch `indices` into the .grad views;
        loss_sums[3] accumulates the critic scale of every call.  index_row (int64 [1] on the device): `indices` is then a
        table [rows, B]; the call uses row index_row[0] and advances it (so a captured graph can be replayed per step).  All tensors float32
        CUDA and contiguous; state [L, D]; action/logprob/adv/r_sum [L]; indices int64 [B]; critic_scale float32 [1]
        is WRITTEN with 1/(r_sum[indices].std()+1e-5), the factor applied to the critic's gradients (agent.py:652)."""
        B = indices.numel() if index_row is None else indices.shape[-1]
        assert B <= self.max_batch and indices.dtype == torch.int64 and indices.is_contiguous()
        if self._structs is None:
            self._build_structs()
        actor, critic, _ = self._structs
        batch = native.PpoBatch(state=state.data_ptr(), action=action.data_ptr(), logprob=logprob.data_ptr(),
                                adv=adv.data_ptr(), r_sum=r_sum.data_ptr(), indices=indices.data_ptr(), B=B,
                                flags=native.PPO_OVERWRITE_GRADS if overwrite else 0,
                                index_row=index_row.data_ptr() if index_row is not None else None,
                                # data parallel: critic gradient left unscaled, target moments written behind the flat gradients
                                dp_moments=self.dp_moments.data_ptr() if defer_critic_scale else None)
        with torch.cuda.device(self.device):
            if adam is not None:   # gradients AND the Adam step, fused into the slab reduction (pime_ppo_minibatch_step)
                assert adam.param is self.flat_param and adam.grad is self.flat_grad
                opt = native.Adam(param=adam.param.data_ptr(), grad=adam.grad.data_ptr(), exp_avg=adam.exp_avg.data_ptr(),
                                  exp_avg_sq=adam.exp_avg_sq.data_ptr(), step=adam.step_count.data_ptr(), n=adam.param.numel(),
                                  lr=adam.lr, beta1=adam.betas[0], beta2=adam.betas[1], eps=adam.eps)
                imap = self.image_map()
                if imap is not None:
                    opt.image_map = imap.data_ptr()
                native.check(native.lib().pime_ppo_minibatch_step(C.byref(actor), C.byref(critic), C.byref(batch),
                                                                  C.c_float(ratio_clip), C.c_float(lambda_entropy),
                                                                  native.ptr(critic_scale), native.ptr(self.moments),
                                                                  native.ptr(self.loss_sums), C.byref(opt), _stream(state)),
                             "pime_ppo_minibatch_step")
                return
            native.check(native.lib().pime_ppo_minibatch_grad(C.byref(actor), C.byref(critic), C.byref(batch),
                                                              C.c_float(ratio_clip), C.c_float(lambda_entropy),
                                                              native.ptr(critic_scale), native.ptr(self.moments),
                                                              native.ptr(self.loss_sums),
                                                              _stream(state)), "pime_ppo_minibatch_grad")


class FlatAdam:
    """torch.optim.Adam semantics (betas 0.9/0.999, eps 1e-8, no weight decay, no amsgrad) on one flat float32
    tensor, as one HIP launch whose step counter lives on the device (graph-replayable)."""

    def __init__(self, param, grad, lr, betas=(0.9, 0.999), eps=1e-8):
        _need_cuda(param)
        self.param, self.grad, self.lr, self.betas, self.eps = param, grad, float(lr), betas, eps
        self.exp_avg = torch.zeros_like(param)
        self.exp_avg_sq = torch.zeros_like(param)
        self.step_count = torch.zeros(2, dtype=torch.float32, device=param.device)   # [0] step, [1] arrival counter (scratch)
        self.param_groups = [{"params": [param], "lr": self.lr}]

    def step(self, images=None, dp=None):
        """images: the FusedPPOGrad whose packed images should follow the step (its image map): the launch then also writes
        every new parameter value into them (pime_adam_step_images) and the caller skips repack().
        dp = (FusedPPOGrad, world): the step of a data-parallel rank behind the all-reduce of fused.flat_grad_dp -- the critic's
        gradient is first scaled by 1 / (std of the union minibatch's targets + 1e-5) (pime_adam_step_dp)."""
        imap = images.image_map() if images is not None else None
        if dp is not None:
            fused, world = dp
            assert fused.flat_param is self.param and fused.flat_grad.data_ptr() == self.grad.data_ptr()
            if fused._structs is None:
                fused._build_structs()
            actor, critic, _ = fused._structs
            opt = native.Adam(param=self.param.data_ptr(), grad=self.grad.data_ptr(), exp_avg=self.exp_avg.data_ptr(),
                              exp_avg_sq=self.exp_avg_sq.data_ptr(), step=self.step_count.data_ptr(), n=self.param.numel(),
                              lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                              image_map=imap.data_ptr() if imap is not None else None,
                              dp_moments=fused.dp_moments.data_ptr(), critic_offset=int(fused.critic_offset), dp_world=int(world))
            with torch.cuda.device(self.param.device):
                native.check(native.lib().pime_adam_step_dp(C.byref(opt), C.byref(actor), C.byref(critic), _stream(self.param)),
                             "pime_adam_step_dp")
            return
        if imap is not None:
            assert images.flat_param is self.param
            if images._structs is None:
                images._build_structs()
            actor, critic, _ = images._structs
            opt = native.Adam(param=self.param.data_ptr(), grad=self.grad.data_ptr(), exp_avg=self.exp_avg.data_ptr(),
                              exp_avg_sq=self.exp_avg_sq.data_ptr(), step=self.step_count.data_ptr(), n=self.param.numel(),
                              lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, image_map=imap.data_ptr())
            with torch.cuda.device(self.param.device):
                native.check(native.lib().pime_adam_step_images(C.byref(opt), C.byref(actor), C.byref(critic), _stream(self.param)),
                             "pime_adam_step_images")
            return
        with torch.cuda.device(self.param.device):
            native.check(native.lib().pime_adam_step(native.ptr(self.param), native.ptr(self.grad), native.ptr(self.exp_avg),
                                                     native.ptr(self.exp_avg_sq), self.param.numel(), C.c_float(self.lr),
                                                     C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                                     native.ptr(self.step_count), _stream(self.param)), "pime_adam_step")

    def zero_grad(self, set_to_none=False):
        self.grad.zero_()

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count, "lr": self.lr}


_TD3_ACTOR_PARAMS = ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias", "net.6.weight", "net.6.bias"]
_TD3_CRITIC_PARAMS = ["net_sa.0.weight", "net_sa.0.bias", "net_sa.2.weight", "net_sa.2.bias", "net_q1.weight", "net_q1.bias",
                      "net_q2.weight", "net_q2.bias"]


class FusedTD3:
    """One TD3 optimizer step -- critic objective, its gradients, Adam, delayed soft update, actor objective through the target
    critic, its gradients, Adam, delayed soft update (elegantrl/agent.py:314-331 of the reference) -- as four HIP launches
    (csrc/td3_fused.hip, `pime_td3_step`).

    The four nets (actor, actor target, critic, critic target) are re-homed into ONE flat float32 tensor each, in the order and at
    the 16-byte-aligned offsets the kernels read them at (`pime_td3_param_offsets`); every nn.Parameter becomes a view into its
    net's flat tensor and every online parameter's .grad a view into the flat gradient the step writes.  The kernels read the
    nn.Linear tensors themselves: there are no packed copies to keep in step with the weights."""

    def __init__(self, act, act_target, cri, cri_target, max_batch, lr, betas=(0.9, 0.999), eps=1e-8):
        dev = next(cri.parameters()).device
        _need_cuda(next(cri.parameters()))
        L = native.lib()
        self.device = dev
        self.D, self.md = int(act.state_dim), int(act.net[0].out_features)
        if not self.supported(act, cri):
            raise native.PimeError(f"fused TD3 step unsupported for state_dim {self.D} width {self.md}: {native.last_error()}")
        self.max_batch = int(max_batch)
        self.nets = (act, act_target, cri, cri_target)
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        f32 = dict(dtype=torch.float32, device=dev)

        def rehome(module, names, which):
            n = L.pime_td3_param_floats(which, self.D, self.md)
            offs = (C.c_int32 * 8)()
            native.check(L.pime_td3_param_offsets(which, self.D, self.md, offs), "pime_td3_param_offsets")
            flat = torch.zeros(n, **f32)
            sd = dict(module.named_parameters())
            with torch.no_grad():
                for name, off in zip(names, offs):
                    p = sd[name]
                    assert p.dtype == torch.float32
                    flat[off:off + p.numel()].copy_(p.detach().reshape(-1))
                    p.data = flat[off:off + p.numel()].view_as(p)
            return flat, list(offs)

        self.act_flat, self.act_off = rehome(act, _TD3_ACTOR_PARAMS, 0)
        self.act_t_flat, _ = rehome(act_target, _TD3_ACTOR_PARAMS, 0)
        self.cri_flat, self.cri_off = rehome(cri, _TD3_CRITIC_PARAMS, 1)
        self.cri_t_flat, _ = rehome(cri_target, _TD3_CRITIC_PARAMS, 1)
        self.act_grad, self.cri_grad = torch.zeros_like(self.act_flat), torch.zeros_like(self.cri_flat)
        for module, names, offs, g in ((act, _TD3_ACTOR_PARAMS, self.act_off, self.act_grad),
                                      (cri, _TD3_CRITIC_PARAMS, self.cri_off, self.cri_grad)):
            sd = dict(module.named_parameters())
            for name, off in zip(names, offs):
                p = sd[name]
                p.grad = g[off:off + p.numel()].view_as(p)
        self.state = {k: torch.zeros_like(t) for k, t in (("act_m", self.act_flat), ("act_v", self.act_flat),
                                                          ("cri_m", self.cri_flat), ("cri_v", self.cri_flat))}
        self.steps_done = torch.zeros(1, **f32)   # optimizer steps applied before table row 0 of the running update (both nets step together)
        self.workspace = None
        self.ensure_batch(self.max_batch)
        self.loss = torch.zeros(4, **f32)          # [0] sum of obj_actor, [1] sum of obj_critic, [2], [3] the last step's
        self.epoch = torch.zeros(1, dtype=torch.int64, device=dev)    # added to the noise epoch: bumped once per update
        self.row = 0                                                   # table row of the next step (host side: a launch argument)
        self._structs()

    def step_dp(self, all_reduce_mean, *args, row=None, **kw):
        """One optimizer step of a data-parallel rank: critic gradients + slab reduction, all-reduce (mean) of the critic's gradient,
        critic Adam + actor gradients + slab reduction, all-reduce of the actor's gradient, actor Adam -- five launches and two
        collectives.  G ranks with their own minibatches make the step of one rank on the union minibatch."""
        r = self.row if row is None else int(row)
        self.step(*args, phases=1 | 16, row=r, **kw)
        all_reduce_mean(self.cri_grad)
        self.step(*args, phases=32 | 4 | 64, row=r, **kw)
        all_reduce_mean(self.act_grad)
        self.step(*args, phases=128, row=r, **kw)
        if row is None:
            self.row += 1

    def begin_update(self):
        """Table row 0 again: the steps of the previous update move into the optimizers' step base, the noise epoch advances."""
        if self.row:
            self.steps_done += float(self.row)
        self.row = 0
        self.epoch += 1

    def ensure_batch(self, batch):
        """Workspace (one partial-gradient slab per workgroup) for minibatches of up to `batch` rows; the optimizer state stays."""
        if self.workspace is None or batch > self.max_batch:
            self.max_batch = max(int(batch), self.max_batch)
            n_ws = native.lib().pime_td3_workspace_floats(self.D, self.md, self.max_batch)
            if n_ws <= 0:
                raise native.PimeError(f"fused TD3 step: {native.last_error()}")
            self.workspace = torch.empty(n_ws, dtype=torch.float32, device=self.device)

    @staticmethod
    def supported(act, cri):
        try:
            if type(act).__name__ != "Actor" or type(cri).__name__ != "CriticTwin":
                return False
            md = act.net[0].out_features
            if cri.net_sa[0].out_features != md or cri.net_sa[0].in_features != act.state_dim + 1 or len(act.net) != 7:
                return False
            return bool(native.lib().pime_td3_supported(int(act.state_dim), int(getattr(act, "action_dim", 1)), int(md))) \
                and next(cri.parameters()).is_cuda
        except Exception:
            return False

    def wraps(self, agent):
        return (agent.act, agent.act_target, agent.cri, agent.cri_target) == self.nets and \
            agent.act.net[0].weight.data_ptr() == self.act_flat.data_ptr()

    def _structs(self):
        s = self.state
        self._actor = native.Td3Net(param=self.act_flat.data_ptr(), target=self.act_t_flat.data_ptr(), grad=self.act_grad.data_ptr(),
                                    exp_avg=s["act_m"].data_ptr(), exp_avg_sq=s["act_v"].data_ptr(), step=self.steps_done.data_ptr(),
                                    lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps)
        self._critic = native.Td3Net(param=self.cri_flat.data_ptr(), target=self.cri_t_flat.data_ptr(), grad=self.cri_grad.data_ptr(),
                                     exp_avg=s["cri_m"].data_ptr(), exp_avg_sq=s["cri_v"].data_ptr(), step=self.steps_done.data_ptr(),
                                     lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps)

    def step(self, buf_state, buf_other, idx, nxt, noise, tau, update_freq, policy_noise, noise_clip=0.5, noise_seed=0,
             noise_epoch=0, soft_mode=2, phases=15, row=None):
        """One optimizer step on table row `row` (default: self.row, which then advances) of idx / nxt (int64 [rows, B]) and noise
        (float32 [rows, B] or None: Philox in the kernel).  The row is a launch argument: a captured graph of an update's steps
        bakes each step's row into its nodes.  phases: bit 0 critic gradients, 1 critic apply, 2 actor gradients, 3 actor apply
        (include/pime_hip.h: what may run beside what); data-parallel callers split an apply around their all-reduce of cri_grad /
        act_grad: 16 / 64 = slab reduction only, 32 / 128 = Adam (+ soft update) from the gradient tensor (step_dp)."""
        _need_cuda(buf_state, buf_other, idx, nxt)
        B = idx.shape[-1]
        advance = row is None
        row = self.row if advance else int(row)
        assert idx.dim() == 2 and 0 <= row < idx.shape[0], (row, idx.shape)
        assert B <= self.max_batch and idx.dtype == nxt.dtype == torch.int64 and idx.is_contiguous() and nxt.is_contiguous()
        assert buf_state.dtype == buf_other.dtype == torch.float32 and buf_state.is_contiguous() and buf_other.is_contiguous()
        assert buf_state.shape[1] == self.D and buf_other.shape[1] == 3
        assert noise is None or (noise.dtype == torch.float32 and noise.is_contiguous() and noise.shape == idx.shape)
        batch = native.Td3Batch(state=buf_state.data_ptr(), other=buf_other.data_ptr(), idx=idx.data_ptr(), nxt=nxt.data_ptr(),
                                noise=noise.data_ptr() if noise is not None else None, row=row, epoch=self.epoch.data_ptr(), B=B,
                                noise_seed=int(noise_seed), noise_epoch=int(noise_epoch), policy_noise=float(policy_noise),
                                noise_clip=float(noise_clip))
        with torch.cuda.device(self.device):
            native.check(native.lib().pime_td3_step(self.D, self.md, C.byref(self._actor), C.byref(self._critic), C.byref(batch),
                                                    C.c_float(tau), int(update_freq), int(soft_mode), int(phases),
                                                    native.ptr(self.workspace), native.ptr(self.loss), _stream(buf_state)),
                         "pime_td3_step")
        if advance and phases & (8 | 128):
            self.row += 1
