"""Data-parallel sharding over the GPUs of one node: one process per GPU, env lanes split into contiguous
slices (rank g owns global lanes [g*N, (g+1)*N)), no communication during rollout, and per optimizer step ONE
all-reduce of a flat gradient buffer over RCCL/xGMI (backend "nccl" on ROCm) -- or gloo in the CPU tests.

The reference has no distributed path at all (SURVEY.md §2 "Native / CUDA / collective inventory: empty");
this is new.  Messages are tiny (67 459 floats = 270 KB for the pH nets at width 128), i.e. latency-bound, so
everything is flattened into one buffer and reduced in one call.
"""
import os

import torch
import torch.distributed as td


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend=None, device=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns a DataParallel or None when
    WORLD_SIZE == 1."""
    rank, world, local = env_rank_world()
    if world <= 1 and os.environ.get("PIME_FORCE_DP") != "1":  # PIME_FORCE_DP=1: exercise the collectives on one rank
        return None
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver only supports dmabuf IPC
        torch.cuda.set_device(local)
    if not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        td.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return DataParallel(rank, world, local, device)


def broadcast_scalar(value, src=0):
    """`value` of rank `src` on every rank (identity when torch.distributed is not initialised)."""
    if not (td.is_available() and td.is_initialized()):
        return value
    dev = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    td.broadcast(t, src=src)
    return float(t.item())


class OneShotAllReduce:
    """The hand-written one-shot all-reduce (mean) of a flat float32 device buffer over peer-mapped memory
    (csrc/allreduce.hip, `pime_oneshot_*`): every rank writes its vector into every peer's inbox, raises a flag, waits for its
    peers' flags and sums the rows in rank order -- one launch on the current stream, HIP-graph capturable, bit-identical on
    every rank.  The 64-byte IPC handles are exchanged once through the process group that is already up."""

    def __init__(self, rank, world, n_floats, device):
        import ctypes as C

        from . import native
        self._lib, self.n = native.lib(), int(n_floats)
        self.device = torch.device(device)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only supports dmabuf IPC
        self._h = C.c_void_p(self._lib.pime_oneshot_create(rank, world, self.n, self.device.index or 0))
        if not self._h:
            raise native.PimeError(f"pime_oneshot_create failed: {native.last_error()}")
        mine = (C.c_ubyte * 64)()
        native.check(self._lib.pime_oneshot_export(self._h, mine), "pime_oneshot_export")
        info = (C.c_int32 * 5)()   # after the export: it may have re-allocated the region as coarse-grained memory
        native.check(self._lib.pime_oneshot_info(self._h, info), "pime_oneshot_info")
        dev = self.device if td.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor(list(mine), dtype=torch.uint8, device=dev)
        gathered = [torch.empty_like(t) for _ in range(world)]
        td.all_gather(gathered, t)
        # which memory every rank obtained and which GPU it sits on (PCI domain / bus / device + host name hash): coarse-grained
        # regions are coherent only between processes that share ONE device, so a fallback anywhere with ranks on different devices
        # is refused -- on every rank alike, from the gathered facts -- and the caller falls back to the RCCL / gloo all-reduce
        import socket
        import zlib
        me = torch.tensor([int(info[0]), int(info[2]), int(info[3]), int(info[4]), zlib.crc32(socket.gethostname().encode()) & 0x7FFFFFFF],
                          dtype=torch.int64, device=dev)
        infos = [torch.empty_like(me) for _ in range(world)]
        td.all_gather(infos, me)
        infos = [i.cpu().tolist() for i in infos]
        self.fine_grained = all(i[0] == 1 for i in infos)
        self.same_device = all(i[1:] == infos[0][1:] for i in infos)
        if not self.fine_grained and not self.same_device:
            self.close()
            raise native.PimeError("one-shot all-reduce refused: the runtime gave coarse-grained memory on rank(s) "
                                   f"{[r for r, i in enumerate(infos) if i[0] != 1]} and the ranks sit on different devices "
                                   "(peer writes would not be coherent while the kernels run)")
        handles = (C.c_ubyte * (64 * world))(*[int(b) for g in gathered for b in g.cpu().tolist()])
        native.check(self._lib.pime_oneshot_connect(self._h, handles), "pime_oneshot_connect")
        td.barrier()   # every rank has mapped every region before the first push

    def __call__(self, flat):
        from . import native
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and flat.numel() == self.n
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        native.check(self._lib.pime_oneshot_allreduce_mean(self._h, flat.data_ptr(), stream), "pime_oneshot_allreduce_mean")
        return flat

    def status(self):
        """0: every call so far completed; non-zero: a peer did not arrive within the kernel's spin limit (synchronises)."""
        return int(self._lib.pime_oneshot_status(self._h))

    def check(self):
        """A timed-out call left the buffer partly or wholly un-averaged and the replicas diverged: fatal, on whichever rank sees it
        (its peers time out on its absence in turn)."""
        from . import native
        st = self.status()
        if st != 0:
            what = "a peer's rows did not arrive within the kernel's spin limit (~2 s)" if st == 1 else "the local grid barrier timed out"
            raise native.PimeError(f"one-shot all-reduce failed (status {st}): {what}; the gradient was NOT averaged and the "
                                   "replicas have diverged -- aborting")

    def close(self):
        if getattr(self, "_h", None):
            from . import native
            native.destroy_handle(self._lib.pime_oneshot_destroy, self._h)   # parked, not freed, while a stream capture is open
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass


class DataParallel:
    def __init__(self, rank, world, local_rank=0, device=None):
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.device = device
        self._flat = None
        # PIME_ONESHOT_ALLREDUCE=1: float32 device buffers are averaged by the hand-written one-shot kernel instead of RCCL.
        # Opt-in: it is validated with two processes on ONE device (tests/test_gpu_oneshot_allreduce.py); visibility of
        # fine-grained memory ACROSS devices over xGMI cannot be exercised on this pool's one-GPU boxes.
        self.use_oneshot = os.environ.get("PIME_ONESHOT_ALLREDUCE") == "1"
        self._oneshot = {}
        # RCCL collectives are kernels on the current stream and can be captured into a HIP graph, and so is the one-shot kernel;
        # gloo's are host calls (all_reduce_mean clears the flag again if the one-shot path has to be refused under gloo)
        self.graph_capturable = td.is_initialized() and (td.get_backend() == "nccl" or self.use_oneshot)

    def lane_offset(self, lanes_per_rank):
        """Global id of this rank's lane 0 (the env kernels' Philox counter word / Mt19937 seed offset)."""
        return self.rank * lanes_per_rank

    def broadcast_module(self, *modules):
        """Make every replica start from rank 0's weights."""
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                td.broadcast(t.data, src=0)

    def all_reduce_sum(self, t):
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return t

    def all_reduce_mean(self, t):
        """Mean over the ranks in ONE collective: RCCL's AVG op on the GPU (no separate divide launch); gloo has no
        AVG, so the CPU tests sum and divide."""
        if self.use_oneshot and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and self.world <= 8:
            ar = self._oneshot.get(t.numel())
            if ar is None:
                from .native import PimeError
                try:
                    ar = self._oneshot[t.numel()] = OneShotAllReduce(self.rank, self.world, t.numel(), t.device)
                except PimeError as exc:   # decided from all-gathered facts: every rank lands here together
                    print(f"| {exc}; using the {td.get_backend()} all-reduce")
                    self.use_oneshot = False
                    self.graph_capturable = td.get_backend() == "nccl"
            if ar is not None:
                return ar(t)
        if td.get_backend() == "nccl":
            td.all_reduce(t, op=td.ReduceOp.AVG)
        else:
            td.all_reduce(t, op=td.ReduceOp.SUM)
            t.div_(self.world)
        return t

    def average_gradients(self, params, extra=None):
        """Flatten every gradient into ONE buffer, all-reduce it once, scatter the mean back.  extra: a small float vector that
        rides behind the gradients in the same collective (the minibatch's target moments); its rank mean is returned."""
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return None
        if extra is not None:
            grads = grads + [extra.to(device=grads[0].device, dtype=grads[0].dtype)]
        n = sum(g.numel() for g in grads)
        key = (n, grads[0].device, grads[0].dtype)   # one staging buffer per gradient set (TD3 alternates critic / actor)
        if not isinstance(self._flat, dict):
            self._flat = {}
        flat = self._flat.get(key)
        if flat is None:
            flat = self._flat[key] = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
        torch.cat([g.reshape(-1) for g in grads], out=flat)
        td.all_reduce(flat, op=td.ReduceOp.SUM)
        flat.div_(self.world)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        return grads[-1] if extra is not None else None

    def check(self):
        """Raises if a one-shot all-reduce of this rank timed out (called where the update synchronises anyway)."""
        for ar in self._oneshot.values():
            ar.check()

    def barrier(self):
        td.barrier()

    def max_over_ranks(self, value):
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return float(t.item())
