"""Device memory released while a stream capture is open (VERDICT r03 task 4).

hipFree under an open stream capture aborts the process; Python finalises handles whenever a reference count drops or its cyclic
collector runs -- round 3 lost a test process that way (gpurun_out/r03b_pytest.log: an env handle garbage-collected inside
AgentTD3's torch.cuda.graph).  The library now parks such memory (pime_capture_begin / _end / _leave, csrc/abi.hip release_device)
and frees it at its next entry point outside the capture."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _env(n=64):
    from pime_amd import gym_control
    return gym_control.make_vec(gym_control.WT_INTEGRATOR, n, device=DEV, seed=1, reward_type="distance")


def test_destroy_is_deferred_while_the_capture_flag_is_set_and_drained_after():
    """No real capture: set the flag, destroy a handle, the free must be parked; clearing the flag drains the queue."""
    import pime_amd.native as nt
    L = nt.lib()
    env = _env()
    env.reset()
    torch.cuda.synchronize()
    assert L.pime_deferred_releases() == 0
    L.pime_capture_begin()
    try:
        env.close()
        assert L.pime_deferred_releases() == 1, "pime_env_destroy under the capture flag must park the slab, not free it"
        L.pime_capture_begin()      # nested captures: the inner end must not drain
        L.pime_capture_end()
        assert L.pime_deferred_releases() == 1
    finally:
        L.pime_capture_end()
    assert L.pime_deferred_releases() == 0, "leaving the outermost capture must drain the parked frees"


def test_env_handle_dropped_inside_a_guarded_capture():
    """The deterministic form of round 3's abort: the last reference to a VecControlEnv goes away (refcount AND a forced cyclic
    collection) inside torch.cuda.graph under native.capture_guard: the process survives, the graph replays, the slab is freed
    when the guard exits."""
    import pime_amd.native as nt
    L = nt.lib()
    env = _env()
    env.reset()
    x = torch.zeros(1024, device=DEV)
    cyc = _Cycle()                  # owns a second handle and sits in a reference cycle: only the cyclic collector finalises it
    cyc.me = cyc
    torch.cuda.synchronize()
    gc.collect()
    g = torch.cuda.CUDAGraph()
    with nt.capture_guard(), torch.cuda.graph(g, capture_error_mode="thread_local"):
        x.add_(1.0)
        del env                     # refcount-driven finalisation inside the capture
        assert L.pime_deferred_releases() == 1
        del cyc                     # the collector is switched off inside the guard: force a collection
        gc.collect()
        assert L.pime_deferred_releases() == 2
        x.add_(1.0)
    assert L.pime_deferred_releases() == 0
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert float(x[0]) == 4.0


class _Cycle:
    def __init__(self):
        self.env = _env(32)
        self.env.reset()
        self.me = None


def test_env_handle_dropped_inside_a_capture_nobody_announced():
    """User code that captures without the guard: the finaliser sees the capture through torch, parks the slab and leaves it parked
    (draining at once would be the hipFree under capture); the next entry point outside the capture frees it."""
    import pime_amd.native as nt
    L = nt.lib()
    env = _env()
    env.reset()
    x = torch.zeros(16, device=DEV)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        x.add_(1.0)
        del env
        assert L.pime_deferred_releases() == 1
    assert L.pime_deferred_releases() == 1, "nothing may drain the queue from inside the capture"
    g.replay()
    torch.cuda.synchronize()
    other = _env()                  # pime_env_create drains
    assert L.pime_deferred_releases() == 0
    other.close()
