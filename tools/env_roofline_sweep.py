#!/usr/bin/env python3
"""Env-kernel HBM roofline sweep: step-per-launch throughput of the fused-residual pH and water-tank
step kernels from 16 384 lanes (the headline config: launch-latency bound, ~1.5 MB per launch) up to 4 M lanes
(bandwidth bound).  Prints one JSON line per (env, N).  Algorithmic bytes per env-step are derived in DESIGN.md §4."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pime_amd import gym_control  # noqa: E402

BYTES = {"ph": 97, "wt": 93, "ph16": 79}   # ph16: state_mode "mixed16" (binary16 I / obs / reward), DESIGN.md §4
PEAK = 8000.0

for kind, env_id, kw in (("ph", gym_control.PH_V35, {}), ("wt", gym_control.WT_INTEGRATOR, dict(reward_type="distance")),
                         ("ph16", gym_control.PH_V35, {})):
    for n in (16384, 65536, 262144, 1 << 20, 1 << 22):
        half = kind == "ph16"
        env = gym_control.make_vec(env_id, n, device="cuda:0", state_mode="mixed16" if half else "mixed", seed=0, **kw)
        if half:
            env.step_residual = env.step_residual_h
        obs_a = (env.reset_h() if half else env.reset()).clone()
        obs_b = torch.empty_like(obs_a)
        a_pre = torch.randn(n, device="cuda:0") * 0.6
        for _ in range(5):
            env.step_residual(a_pre, obs_a, out_obs=obs_b)
            obs_a, obs_b = obs_b, obs_a
        # the host-side mirror bookkeeping of vec_env is outside the kernel: time the launches with events
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps):
            env.step_residual(a_pre, obs_a, out_obs=obs_b)
            obs_a, obs_b = obs_b, obs_a
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        gbs = BYTES[kind] * n / (ms * 1e-3) / 1e9
        print(json.dumps({"env": kind, "lanes": n, "us_per_step_launch": ms * 1e3, "env_steps_per_s": n / (ms * 1e-3),
                          "algorithmic_GBps": gbs, "frac_of_8TBps": gbs / PEAK}), flush=True)
        env.close()
        del env
        torch.cuda.empty_cache()
